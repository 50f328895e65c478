#!/usr/bin/env python3
"""bench.py — training patches/s of the dual-modal fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): synthetic 145x145 scene, 200-band HSI + 1-band SAR, 11x11
patches, 17 logits, batch 256 per GPU, fp32.  N > 1 uses the same scene with per-GPU batch 256 (weak scaling),
one flat fp32 gradient all-reduce per step over RCCL.  A "step" = forward + CE + backward + Adam on one batch;
inputs (padded scenes, the shuffled coordinate/label plan) are resident in HBM before the timed region.

One JSON line on rank 0.  `value` = N*B*K / (max-over-ranks wall time of exactly K steps, barrier +
synchronize on both sides).  `roofline` is for the dominant kernel (the fused per-patch fwd+CE+bwd kernel):
algorithmic bytes per launch (B x 194,568 B) / its mean duration, measured here with HIP events on
the launch stream in an instrumented pass of the same steps.  `cpu_baseline` = the CPU oracle's restatement of
the reference train loop (oracle/solver_ref.py) timed on this host on a bounded number of the same steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT]

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)


def make_cfg(args):
    from dmf.synth import class_colors
    K = args.classes + 1
    return {'patch_size': args.patch, 'Categories_Number': K, 'data_city': 'syn',
            'DATA_DICT': {'syn': {'size': [args.size, args.size, args.bands], 'color': class_colors(K)}},
            'scale': 1, 'aux_bands': args.aux_bands,
            'gmf': {'width': args.width, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}


def build_problem(args, cfg):
    from dmf import synth
    from function.function import data_padding, data_padding_aux, split_data_old
    primary, aux, label = synth.make_scene(args.size, args.size, args.bands, args.aux_bands, 1, n_classes=args.classes, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        xyl, idx = split_data_old(label, cfg)
    labelled = np.array(idx[1])
    g = torch.Generator().manual_seed(3407)                       # test.py:8
    perm = torch.randperm(len(labelled), generator=g).numpy()
    n_train = int(args.train_rate * len(labelled))
    train, test = labelled[perm[:n_train]], labelled[perm[n_train:]]
    xy = np.concatenate([xyl[0], xyl[1]], 1).astype(np.int32)
    lab = xyl[2].reshape(-1).astype(np.int32)
    return MS, PAN, xy, lab, train, test


def make_plan(train, n_steps, B, seed):
    """Concatenated per-epoch permutations of the train pixels, cut into n_steps batches of B."""
    g = torch.Generator().manual_seed(seed)
    out = []
    need = n_steps * B
    while sum(len(o) for o in out) < need:
        out.append(train[torch.randperm(len(train), generator=g).numpy()])
    return np.concatenate(out)[:need]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--size', type=int, default=145)
    ap.add_argument('--bands', type=int, default=200)
    ap.add_argument('--aux-bands', type=int, default=1)
    ap.add_argument('--width', type=int, default=40, help='gmf.width: feature channels per branch (40 or 32)')
    ap.add_argument('--patch', type=int, default=11)
    ap.add_argument('--classes', type=int, default=16)
    ap.add_argument('--train-rate', type=float, default=0.10)
    ap.add_argument('--steps-per-graph', type=int, default=50, help='0 = eager launches')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='CPU baseline budget (rank 0, N=1 only)')
    ap.add_argument('--no-cpu', action='store_true')
    args = ap.parse_args()
    if args.bands == 224 and args.width == 40:
        args.width = 32                     # the compiled 224-band instance

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d'
                             % (args.gpus, args.gpus))
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on cuda:0, gloo instead of RCCL
    if os.environ.get('DMF_SINGLE_DEVICE') == '1':
        local_rank = 0
    backend = os.environ.get('DMF_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    from dmf import lib
    from dmf.engine import EvalEngine, Scene, TrainEngine
    from model.gmfnet import Net

    cfg = make_cfg(args)
    MS, PAN, xy_tab, lab_tab, train, test = build_problem(args, cfg)
    B, K_steps, W_steps = args.batch, args.steps, args.warmup
    torch.manual_seed(3407)
    net = Net(cfg).to(dev)
    init_state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    scene = Scene(MS, PAN, dev)
    comm = None
    if world > 1 and os.environ.get('DMF_ALLREDUCE', 'xgmi') == 'xgmi':
        from dmf import xgmi
        comm = xgmi.create(sum(p.numel() for p in net.parameters()), pg,     # None on every rank if it cannot be proven
                           timeout_ms=int(os.environ.get('DMF_XGMI_TIMEOUT_MS', 20000)))
    eng = TrainEngine(net, scene, B, lr=1e-3, process_group=pg, comm=comm)

    total = W_steps + K_steps
    plan_idx = make_plan(train, total, B * world, seed=1)                 # global batches of world*B
    mine = plan_idx.reshape(total, world, B)[:, rank, :].reshape(-1)      # this rank's contiguous shard per step
    eng.load_plan(xy_tab[mine], lab_tab[mine])
    spg = args.steps_per_graph if (world == 1 or comm is not None) else 0   # RCCL path: eager launches

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up (W steps; also builds the graph), then EXACTLY K timed steps
    eng.run_plan(W_steps, 0)
    if comm is not None:
        # the exchange has now run W real steps: if any rank saw a wait time out, every rank drops to the RCCL
        # all-reduce and repeats the warm-up from the initial weights (decided together, before anything is timed)
        import torch.distributed as dist
        bad = torch.tensor([comm.status()], dtype=torch.int32, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) != 0:
            if rank == 0:
                print('[bench] one-shot exchange timed out during warm-up; falling back to the RCCL all-reduce',
                      file=sys.stderr, flush=True)
            comm = None
            net.load_state_dict(init_state)
            eng = TrainEngine(net, scene, B, lr=1e-3, process_group=pg, comm=None)
            eng.load_plan(xy_tab[mine], lab_tab[mine])
            spg = 0
            eng.run_plan(W_steps, 0)
    if spg:
        eng._capture(spg)                       # capture restores state: no steps are consumed
    sync()
    t0 = time.perf_counter()
    eng.run_plan(K_steps, spg)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if comm is not None and comm.status() != 0:
        raise SystemExit('rank %d: a gradient exchange timed out waiting for a peer — the timing is void' % rank)
    losses = eng.mean_losses().numpy() if world == 1 else np.zeros(0)
    value = world * B * K_steps / dt

    # ---- instrumented pass: mean duration of the dominant kernel, HIP events on the launch stream
    n_inst = min(K_steps, 200)
    P, C, C2 = args.patch, args.bands, args.aux_bands
    alg_patch = 2 * 4 * (P * P * C + P * P * C2)
    kern_ms = None
    if rank == 0 or world > 1:
        inst_plan_xy = torch.from_numpy(xy_tab[mine[:n_inst * B]]).to(dev)
        inst_lab = torch.from_numpy(lab_tab[mine[:n_inst * B]]).to(dev)
        theta2 = eng.theta.clone()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_inst)]
        ws = eng.ws
        torch.cuda.synchronize()
        for i in range(n_inst):
            inp = lib.input_gather(eng.shape, scene.A, scene.B, inst_plan_xy[i * B:(i + 1) * B])
            ev[i][0].record()
            lib.train_fwd_bwd(eng.shape, inp, theta2, net.pool_w, inst_lab[i * B:(i + 1) * B], 1.0 / B, eng.logits, eng.loss, ws)
            ev[i][1].record()
        torch.cuda.synchronize()
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev[n_inst // 10:]]))

    # ---- kappa of the trained net on the held-out split: on-device confusion matrix; with N ranks every rank
    # classifies its shard of the pixels and the K x K matrices are all-reduced
    n_test = min(len(test), 8192)
    ev_eng = EvalEngine(net, scene, 2048)
    m_test = ev_eng.confusion(xy_tab[test[:n_test]], lab_tab[test[:n_test]], process_group=pg).cpu().numpy().astype(np.float64)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
    if rank != 0:
        return

    achieved = B * alg_patch / (kern_ms * 1e-3)
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get('patch_kernel_hbm_bytes_per_launch')
        except Exception:
            traffic = None
    out = {
        'metric': 'training patches/sec + kappa, 11x11x200 HSI + 11x11x1 SAR, 1/2/4/8 MI355X',
        'value': value, 'unit': 'patches/s', 'n_gpus': world, 'steps': K_steps, 'warmup': W_steps,
        'ms_per_step': dt / K_steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: synthetic %dx%d scene, %d-band HSI + %d-band SAR, %dx%d patches, %d logits, '
                               'batch %d per GPU, fused HIP fwd+CE+bwd+Adam'
                               % ('configs[1]' if (C, C2, args.size) == (200, 1, 145) else
                                  ('configs[3] shape' if (C, C2, args.size) == (224, 3, 512) else 'custom'),
                                  args.size, args.size, C, C2, P, P, args.classes + 1, B),
                   'global_batch': B * world, 'parallelism': 'dp%d' % world,
                   'launch': ('hipGraph x%d steps' % spg) if spg else 'eager',
                   'allreduce': 'none' if world == 1 else ('xgmi one-shot, fused in the reduce+Adam launch' if comm is not None
                                                           else 'rccl all_reduce of one flat fp32 gradient')},
        'roofline': {'bound': 'hbm', 'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK, 'traffic': traffic,
                     'kernel': 'dmf::patch_kernel<Shape<%d,%d,%d,1,%d,..>, MODE_TRAIN>' % (C, C2, P, net.arch['F']), 'kernel_ms': kern_ms,
                     'algorithmic_bytes_per_launch': B * alg_patch},
        'step_frac_of_hbm_roof': value / world * (alg_patch + 28.0 * eng.theta.numel() / B) / HBM_PEAK,
    }
    if losses.size:
        out['loss_first_last'] = [float(losses[0]), float(losses[-1])]

    # ---- kappa of the trained net on the held-out split (on-device confusion matrix)
    from indicators.kappa import aa_oa_quiet
    aa, oa, kp = aa_oa_quiet(m_test)
    out['kappa'] = {'gpu': kp, 'oa': oa, 'aa': aa, 'test_patches': int(n_test), 'train_steps': total}

    # ---- CPU baseline: oracle restatement of the reference loop on the same first steps (N == 1 only)
    if world == 1 and not args.no_cpu:
        from oracle.gmfnet_ref import Net as RefNet
        from oracle import solver_ref, datapath_ref
        n_thr = min(len(os.sched_getaffinity(0)), 16)        # the GPU box gives one GPU's job a 16-core share
        torch.set_num_threads(n_thr)
        ref = RefNet(cfg)
        ref.load_state_dict(init_state)
        t_start = time.perf_counter()
        n_cpu = [0]

        class Stop(Exception):
            pass

        def on_step(n):
            n_cpu[0] = n
            if time.perf_counter() - t_start > args.cpu_seconds:
                raise Stop()
        opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
        cpu_losses = []
        try:
            # 25-step chunks so a Stop keeps the losses of finished chunks
            for c0 in range(0, min(total, 4000), 25):
                sl = mine[c0 * B:(c0 + 25) * B]
                l, opt = solver_ref.train_steps(ref, MS, PAN, xy_tab[sl], lab_tab[sl], B, P, 1, optimizer=opt,
                                                on_step=lambda n, c0=c0: on_step(c0 + n))
                cpu_losses += l
        except Stop:
            pass
        t_cpu = time.perf_counter() - t_start
        out['cpu_baseline'] = {'value': n_cpu[0] * B / t_cpu, 'unit': 'patches/s', 'cores': torch.get_num_threads(),
                               'kind': 'port',
                               'sample': 'first %d of the same train steps (batch %d), oracle/solver_ref.py on torch-CPU fp32, %.1f s'
                                         % (n_cpu[0], B, t_cpu)}
        # parity of the first steps' losses (same init, same batches) ...
        n_cmp = min(len(cpu_losses), len(losses))
        if n_cmp:
            d = np.abs(np.array(cpu_losses[:n_cmp]) - losses[:n_cmp])
            out['cpu_baseline']['max_abs_loss_diff_first_20_steps'] = float(d[:20].max())
            out['cpu_baseline']['max_abs_loss_diff_first_%d_steps' % n_cmp] = float(d.max())
        # kappa of the CPU forward path on the GPU-trained weights (same weights, same patches: must be identical)
        n_kt = min(n_test, 2048)
        ref_f = RefNet(cfg)
        ref_f.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
        m_f, _ = solver_ref.evaluate(ref_f, MS, PAN, xy_tab[test[:n_kt]], lab_tab[test[:n_kt]], args.classes + 1, P, 1)
        m_g = ev_eng.confusion(xy_tab[test[:n_kt]], lab_tab[test[:n_kt]]).cpu().numpy().astype(np.float64)
        out['kappa'].update({'same_weights_test_patches': int(n_kt), 'same_weights_gpu': aa_oa_quiet(m_g)[2],
                             'same_weights_cpu_forward': aa_oa_quiet(m_f)[2],
                             'same_weights_predictions_differing': int(np.abs(m_f - m_g).sum() // 2)})
        # ... and kappa after the SAME number of steps on the same held-out patches: CPU net vs a fresh GPU run
        # (two fp32 ADAM trajectories drift apart chaotically once summation orders differ; this is reported, the
        # same-weights figures above are the parity claim)
        n_k = len(cpu_losses)
        if n_k:
            m_cpu, _ = solver_ref.evaluate(ref, MS, PAN, xy_tab[test[:n_kt]], lab_tab[test[:n_kt]], args.classes + 1, P, 1)
            net2 = Net(cfg).to(dev)
            net2.load_state_dict(init_state)
            eng2 = TrainEngine(net2, scene, B, lr=1e-3)
            eng2.load_plan(xy_tab[mine[:n_k * B]], lab_tab[mine[:n_k * B]])
            eng2.run_plan(n_k, 0)
            m_gpu = EvalEngine(net2, scene, 2048).confusion(xy_tab[test[:n_kt]], lab_tab[test[:n_kt]]).cpu().numpy().astype(np.float64)
            out['kappa'].update({'parity_steps': n_k, 'parity_test_patches': int(n_kt),
                                 'cpu_after_parity_steps': aa_oa_quiet(m_cpu)[2], 'gpu_after_parity_steps': aa_oa_quiet(m_gpu)[2],
                                 'confusion_entries_differing': int(np.abs(m_cpu - m_gpu).sum() // 2)})
    print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
