"""GPU: the data-parallel engine path (world_size 2).  Two processes share the one GPU of the test box and talk
over gloo (RCCL refuses two ranks on one device); the code path is the one `bench.py --gpus N` runs over RCCL:
dmf_grad_reduce -> all-reduce(sum) of ONE flat gradient -> dmf_adam_step(grad_scale = 1/N).  After 3 steps the
2-rank parameters must equal those of a single process that trains on the concatenated global batches.

The same check runs over the one-shot xGMI exchange (dmf_grad_reduce_xgmi_adam, HIP-IPC mapped peer buffers) with 2, 3 and
4 ranks, stepping eagerly and replaying a captured hipGraph of the whole data-parallel step; on the one-GPU box the "peers"
are processes on the same device, which exercises the IPC mapping, tagged words, parities and the rank-ordered sum, not the xGMI
links themselves.  The exchange makes a kernel wait for a kernel of ANOTHER process.  With three and more processes on ONE
GPU that dead-locks on residency (measured in round 3: the ranks that reach the exchange first fill every compute unit with
waiting blocks and the last rank's kernels find no unit), so each rank of those cases launches on a stream that owns a
share of the compute units (`dmf.xgmi.cu_share_stream`) — its "own GPU".  A wait that times out all the same is reported as an
EXPECTED FAILURE with that reason (never a pass, never a silent skip); wrong sums always fail.  The 3-rank case is the one
that exposed the reader-side caching of the first (pull) design (csrc/dmf_xgmi.h); it runs against the push design with
the store-acknowledgement wait."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'dual-modal-fusion_amd')
CFG = {'patch_size': 5, 'Categories_Number': 5, 'data_city': 's', 'DATA_DICT': {'s': {'size': [20, 20, 8]}}, 'scale': 1,
       'aux_bands': 1, 'gmf': {'width': 40}}
B, STEPS = 32, 3          # global batch 2*B = 64; xgmi cases: 6 steps of 48 (divisible by 2 and 3)


def _problem(n=2 * B * STEPS):
    sys.path[:0] = [PKG, REPO]
    from dmf import synth
    from function.function import data_padding, data_padding_aux
    primary, aux, label = synth.make_scene(20, 20, 8, 1, 1, n_classes=4, seed=3)
    MS = data_padding(primary, CFG, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, CFG).astype(np.float32)
    rng = np.random.default_rng(0)
    xy = np.stack([rng.integers(0, 20, n), rng.integers(0, 20, n)], 1).astype(np.int32)
    lab = np.maximum(label[xy[:, 0], xy[:, 1]].astype(np.int32), 1)
    return MS, PAN, xy, lab


def _train_xgmi(rank, world, port, q, graph_steps):
    """GB = 48 patches per step, 6 steps; world ranks (or one) — returns parameters and per-step local losses."""
    import torch.distributed as dist
    MS, PAN, xy, lab = _problem(48 * 6)
    from dmf import xgmi
    from dmf.engine import Scene, TrainEngine
    from dmf.parallel import shard_batch
    from model.gmfnet import Net
    GB, NS = 48, 6
    pg = comm = None
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        import datetime
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        pg = dist.group.WORLD
    torch.manual_seed(0)
    net = Net(CFG).to('cuda:0')
    if world > 1:
        comm = xgmi.create(sum(p.numel() for p in net.parameters()), pg, timeout_ms=3000)
        if comm is None:                 # the known-answer rounds did not complete: see _NOT_CORESIDENT below
            if rank == 0:
                q.put('timeout')
            dist.destroy_process_group()
            return
    if world >= 3:                   # every rank on its own compute units, as on a node (why: dmf.xgmi.cu_share_stream)
        torch.cuda.set_stream(xgmi.cu_share_stream(rank, world))
    eng = TrainEngine(net, Scene(MS, PAN, 'cuda:0'), GB // world, lr=1e-2, process_group=pg, comm=comm)
    gxy, glab = xy[:NS * GB].reshape(NS, GB, 2), lab[:NS * GB].reshape(NS, GB)
    lo, hi = shard_batch(GB, rank, world)
    eng.load_plan(gxy[:, lo:hi].reshape(-1, 2), glab[:, lo:hi].reshape(-1))
    if graph_steps:                  # bench.py's sequence: eager warm-up, capture, replay, eager remainder
        eng.run_plan(1, 0)
        eng.run_plan(NS - 1, graph_steps)
    else:
        eng.run_plan(NS, 0)
    torch.cuda.synchronize()
    timed_out = comm is not None and comm.status() != 0
    if comm is not None:
        # the generic small all-reduce on the same communicator, against the group's own all_reduce
        g = torch.Generator().manual_seed(100 + rank)
        for it in range(20):
            v = torch.randn(1000 + 37 * it, generator=g)
            mine = v.cuda()
            comm.allreduce_(mine)
            parts = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(parts, v)
            want = parts[0].clone()
            for r in range(1, world):
                want += parts[r]                       # rank order, like the kernel
            got = mine.cpu()
            timed_out = timed_out or comm.status() != 0      # sticky: later rounds no longer wait, every rank keeps
            if timed_out:                                     # calling the collectives so that nobody is left behind
                continue
            if not torch.equal(got, want):
                bad = (got != want).nonzero().reshape(-1)
                cand = {('-r%d' % r): float((got - (want - parts[r])).abs().max()) for r in range(world)}
                print('rank %d round %d: %d/%d elements differ, first at %s, max|diff| %.3e; diff if source r were missing: %s'
                      % (rank, it, bad.numel(), got.numel(), bad[:5].tolist(), float((got - want).abs().max()), cand), flush=True)
            assert torch.equal(got, want), 'xgmi all-reduce differs from the rank-ordered sum (round %d)' % it
    assert eng.step_count == NS
    if world > 1:                        # a wait that timed out on ANY rank voids the run for all of them
        flag = torch.tensor([1 if timed_out else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        timed_out = bool(flag.item())
    if rank == 0:
        q.put('timeout' if timed_out else eng.theta.cpu().numpy())
    if world > 1:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


_NOT_CORESIDENT = ('a rank waited longer than 3 s for a peer: the per-GPU processes of this test share ONE GPU and their '
                   'kernels were not scheduled side by side this time (on a node every rank has its own GPU); the arithmetic '
                   'of the exchange could not be checked in this run')


def _run_ranks(target, world, extra):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() + 7 * world) % 1000
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    [p.start() for p in procs]
    try:
        out = q.get(timeout=120)
    finally:
        [p.join(60) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return out


@pytest.mark.parametrize('world,graph_steps', [(2, 0), (2, 4), (3, 0), (3, 4), (4, 4)])
def test_xgmi_exchange_dp_equals_single_rank_global_batch(world, graph_steps):
    many = _run_ranks(_train_xgmi, world, (graph_steps,))
    if isinstance(many, str):
        pytest.xfail(_NOT_CORESIDENT)
    one = _run_ranks(_train_xgmi, 1, (0,))
    err = np.abs(many - one).max()
    print('%d-rank xgmi (graph %d) vs 1-rank parameters: max abs diff %.2e' % (world, graph_steps, err))
    assert err < 2e-5


def _train(rank, world, port, q):
    import torch.distributed as dist
    MS, PAN, xy, lab = _problem()
    from dmf.engine import Scene, TrainEngine
    from dmf.parallel import shard_batch
    from model.gmfnet import Net
    pg = None
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        pg = dist.group.WORLD
    torch.manual_seed(0)
    net = Net(CFG).to('cuda:0')
    eng = TrainEngine(net, Scene(MS, PAN, 'cuda:0'), (2 * B) // world, lr=1e-2, process_group=pg)
    gxy, glab = xy.reshape(STEPS, 2 * B, 2), lab.reshape(STEPS, 2 * B)
    lo, hi = shard_batch(2 * B, rank, world)
    eng.load_plan(gxy[:, lo:hi].reshape(-1, 2), glab[:, lo:hi].reshape(-1))
    eng.run_plan(STEPS, 0)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(eng.theta.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_dp_equals_single_rank_global_batch():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_train, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    two = q.get(timeout=300)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    p1 = ctx.Process(target=_train, args=(0, 1, port, q))
    p1.start()
    one = q.get(timeout=300)
    p1.join(120)
    assert p1.exitcode == 0
    err = np.abs(two - one).max()
    print('2-rank vs 1-rank parameters after %d steps: max abs diff %.2e' % (STEPS, err))
    assert err < 2e-5


def _eval_sharded(rank, world, port, q):
    import torch.distributed as dist
    MS, PAN, xy, lab = _problem(301)                       # 301 pixels: shards of 151 and 150
    from dmf.engine import EvalEngine, Scene
    from model.gmfnet import Net
    pg = None
    if world > 1:
        import datetime
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        pg = dist.group.WORLD
    torch.manual_seed(0)
    net = Net(CFG).to('cuda:0')
    ev = EvalEngine(net, Scene(MS, PAN, 'cuda:0'), 64)
    m = ev.confusion(xy, lab, process_group=pg).cpu().numpy()
    lm = ev.label_map(xy, 20, 20, process_group=pg).cpu().numpy()
    if rank == 0:
        q.put((m, lm))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_evaluation_equals_single_rank():
    """Evaluation / colouring over a process group: every rank classifies its shard of the pixels, the
    K x K confusion matrix is all-reduced (sum), label-map tiles are merged — identical to one rank doing all of it."""
    m2, lm2 = _run_ranks(_eval_sharded, 2, ())
    m1, lm1 = _run_ranks(_eval_sharded, 1, ())
    assert m1.sum() == 301 and np.array_equal(m1, m2) and np.array_equal(lm1, lm2)


def _solver_dp(rank, world, port, q, tmp):
    """`Solver(cfg).run()` on the G9 scene, optionally as one of `world` data-parallel ranks (gloo transport)."""
    import json
    import torch.distributed as dist
    sys.path[:0] = [PKG, REPO]
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'g9_trajectory.npz'), allow_pickle=False)
    d = os.path.join(tmp, 'scene') + '/'
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        np.save(d + 'ms4.tif.npy', g['primary']); np.save(d + 'pan.tif.npy', g['aux']); np.save(d + 'label.npy', g['label'])
        os.makedirs(os.path.join(tmp, 'out%d' % world), exist_ok=True)
    cfg = json.loads(str(g['cfg']))
    cfg.update(data_address=d, RESULT_output=os.path.join(tmp, 'out%d' % world) + '/', RESULT_excel=os.path.join(tmp, 'r%d.xlsx' % world),
               nohup=1, device='cuda:0', epoch=6, batchsize=32)
    cfg['test']['full'] = 1
    cfg['color']['index'] = 1
    from solver.mainsolver import Solver
    if world > 1:
        import datetime
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
        dist.barrier()
    torch.manual_seed(3407)
    s = Solver(cfg)
    if world > 1:
        s.process_group, s.rank, s.world = dist.group.WORLD, rank, world
    s.run()
    if rank == 0:
        q.put((s.cur_model.flat_parameters().cpu().numpy(), s.test_matrix, s.label_maps[1]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_solver_data_parallel_equals_single_rank():
    """The solver itself as 2 data-parallel ranks: same shuffled stream, per-rank shards of every batch (the G9 epoch
    is 3 x 32 + 18 patches: the short batch is an even 18), exchange through the process group, sharded test and colour
    passes.  Weights after 6 epochs, the whole-split confusion matrix and the label map must equal the 1-rank run."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix='dmf_soldp_')
    try:
        two = _run_ranks(_solver_dp, 2, (tmp,))
        one = _run_ranks(_solver_dp, 1, (tmp,))
        err = np.abs(two[0] - one[0]).max()
        print('solver 2-rank vs 1-rank parameters after 6 epochs: max abs diff %.2e' % err)
        assert err < 1e-5
        assert np.array_equal(two[1], one[1]) and np.array_equal(two[2], one[2])
    finally:
        shutil.rmtree(tmp)


@pytest.mark.parametrize('world', [3, 4])      # (streams beyond the device's concurrent hardware queues would serialise and wait for ever)
def test_xgmi_protocol_many_ranks_one_process(world):
    """The exchange protocol itself (slots, parities, tagged words, sequence numbers, rank-ordered sum) for more than two ranks,
    made deterministic on a one-GPU box: the `world` ranks are `world` STREAMS of this process (kernels of one process do
    run side by side), each with its own owner-uncached inbox and status block; 12 rounds reuse every parity slot six times and
    the vector length changes from round to round.  What this cannot cover — the IPC mapping and real xGMI links — is what
    the multi-process cases above and bench.py's warm-up admission check are for."""
    sys.path[:0] = [PKG, REPO]
    from dmf import lib
    cap = 5000
    data_bytes, flag_bytes = lib.xgmi_sizes(cap, world)
    bufs = [(lib.xgmi_alloc(data_bytes), lib.xgmi_alloc(flag_bytes)) for _ in range(world)]
    comms = []
    for r in range(world):
        c = lib.XgmiComm(world=world, rank=r, capacity=cap, timeout_ms=3000, seq_bias=0)
        for q in range(world):
            c.data[q], c.flags[q] = bufs[q]
        comms.append(c)
    # Every rank's stream must own a HARDWARE queue: HIP multiplexes the streams of a process onto a few queues
    # (GPU_MAX_HW_QUEUES, 4 by default), and two ranks that share one run their kernels one after the other — the first waits
    # for a peer that is queued behind it, until the time-out.  Which streams share depends on how many streams the process
    # has made before (this test timed out in "round 1" in some orders of the suite and in none of others: round 2's
    # gpurun_out/r2_stage2d.log, round 3's r3b_traj.log).  A stream with a compute-unit mask carries its own queue (the mask is
    # a property of the queue), and the rank's own share of the units on top (dmf.xgmi.cu_share_stream).  Each stream also
    # runs one trivial kernel to completion before the first exchange (queue creation does not happen beside a spinning kernel).
    from dmf import xgmi
    streams = [xgmi.cu_share_stream(r, world) for r in range(world)]
    for st in streams:
        with torch.cuda.stream(st):
            torch.zeros(64, device='cuda').add_(1.0)
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(7)
    try:
        for seq in range(1, 13):
            n = 700 + 331 * (seq % 5)
            vals = [torch.randn(n, generator=g) for _ in range(world)]
            dev = [v.cuda() for v in vals]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    lib.xgmi_allreduce(comms[r], dev[r], n, seq)
            torch.cuda.synchronize()
            # (streams of one process with live hardware queues run side by side: a timed-out wait is a failure)
            assert all(lib.xgmi_status(c) == 0 for c in comms), 'round %d: a rank timed out waiting for a peer' % seq
            want = vals[0].clone()
            for r in range(1, world):
                want += vals[r]                        # rank order, like the kernel
            for r in range(world):
                assert torch.equal(dev[r].cpu(), want), 'rank %d differs from the rank-ordered sum in round %d' % (r, seq)
    finally:
        torch.cuda.synchronize()
        for d, f in bufs:
            lib.xgmi_free(d); lib.xgmi_free(f)


def _train_rccl_one_rank(rank, world, port, q, graph_steps):
    import datetime
    import torch.distributed as dist
    MS, PAN, xy, lab = _problem(48 * 6)
    from dmf.engine import Scene, TrainEngine
    from model.gmfnet import Net
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=torch.device('cuda', 0))
    out = {}
    for forced in (False, True):
        torch.manual_seed(0)
        net = Net(CFG).to('cuda:0')
        eng = TrainEngine(net, Scene(MS, PAN, 'cuda:0'), 48, lr=1e-2, process_group=dist.group.WORLD)
        eng._force_collective = forced         # True: dmf_grad_reduce -> RCCL all-reduce -> dmf_adam_step, as with N ranks
        eng.load_plan(xy, lab)
        if graph_steps:
            assert eng._graphable()
            eng.run_plan(1, 0)
            eng.run_plan(4, graph_steps)
            assert (eng.graph is not None) or not forced or not getattr(eng, '_rccl_graph', True)
            out['captured_%d' % forced] = eng.graph is not None
            eng.run_plan(1, 0)
        else:
            eng.run_plan(6, 0)
        torch.cuda.synchronize()
        out[forced] = (eng.theta.cpu().numpy(), eng.mean_losses().numpy())
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.parametrize('graph_steps', [0, 2])
def test_rccl_form_of_the_step_on_a_one_rank_group(graph_steps):
    """The fallback of the data-parallel step — dmf_grad_reduce, the process group's all-reduce (RCCL), dmf_adam_step — run on
    a ONE-rank NCCL group, eagerly and captured in a hipGraph (TrainEngine._graphable: RCCL collectives are capturable),
    against the fused single-GPU step on the same batches.  One rank is all a one-GPU box can give RCCL; what this covers is
    the process-group plumbing, the capture and replay of the collective inside the step's graph and the unfused ADAM."""
    out = _run_ranks(_train_rccl_one_rank, 1, (graph_steps,))
    (th0, l0), (th1, l1) = out[False], out[True]
    print('RCCL form vs fused step: parameters max abs diff %.2e, losses %.2e; graph captured: %s'
          % (np.abs(th0 - th1).max(), np.abs(l0 - l1).max(), out.get('captured_1')))
    assert np.abs(th0 - th1).max() < 2e-6 and np.abs(l0 - l1).max() < 1e-6
    if graph_steps:
        assert out['captured_1'], 'the all-reduce was not captured in the step graph (eager fallback taken)'


def test_xgmi_wait_is_bounded_and_sticky():
    """A peer that never arrives: the waiting lanes give up after timeout_ms, the communicator's status turns 1 and stays 1,
    and later exchanges on it return without waiting (a stuck peer must cost one time-out, not one per step) — the exit
    condition every wave of the exchange reaches."""
    import time
    sys.path[:0] = [PKG, REPO]
    from dmf import lib
    cap, world = 1000, 2
    data_bytes, flag_bytes = lib.xgmi_sizes(cap, world)
    bufs = [(lib.xgmi_alloc(data_bytes), lib.xgmi_alloc(flag_bytes)) for _ in range(world)]
    c = lib.XgmiComm(world=world, rank=0, capacity=cap, timeout_ms=200, seq_bias=0)
    for q in range(world):
        c.data[q], c.flags[q] = bufs[q]
    try:
        v = torch.ones(cap, device='cuda')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.xgmi_allreduce(c, v, cap, 1)                      # rank 1 never runs
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        assert lib.xgmi_status(c) == 1
        assert 0.15 < t1 - t0 < 2.0, 'the wait took %.3f s for a 0.2 s time-out' % (t1 - t0)
        lib.xgmi_allreduce(c, v, cap, 2)
        torch.cuda.synchronize()
        assert time.perf_counter() - t1 < 0.1 and lib.xgmi_status(c) == 1
    finally:
        torch.cuda.synchronize()
        for d, f in bufs:
            lib.xgmi_free(d); lib.xgmi_free(f)


# ---------------------------------------------------------------------------------------------- stage 2, data parallel
QCFG = {'patch_size': 5, 'Categories_Number': 5, 'data_city': 's', 'DATA_DICT': {'s': {'size': [20, 20, 4]}},
        'gmf': {'width': 40, 'single_input': 1}, 'dqtl': {'alpha': 0.1, 'beta': 0.05, 'gamma': 1.0, 'epsilon': 1e-8, 'tao': 0.1}}


def _train_stage2(rank, world, port, q):
    """4 steps of the stage-2 engine on global batches of 24 pixels; world ranks take 24 / world each.  qua_loss couples the
    whole batch, so the ranks gather their logits and evaluate it on the GLOBAL batch (engine.QuaTrainEngine._global_loss)."""
    import torch.distributed as dist
    sys.path[:0] = [PKG, REPO]
    from dmf import synth
    from dmf.engine import QuaScene, QuaTrainEngine
    from function.function import data_padding
    from model.gmfnet import Net
    pg = None
    if world > 1:
        import datetime
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        pg = dist.group.WORLD
    ms, pan, label = synth.make_scene(20, 20, 4, 1, 1, n_classes=4, seed=5)
    g = np.random.default_rng(2)
    scenes = [data_padding(x, QCFG, 'ms') for x in (ms, ms[::-1].copy(), ms + 0.1 * g.standard_normal(ms.shape), ms * 0.5)]
    GB, NS = 24, 4
    xy = np.stack([g.integers(0, 20, GB * NS), g.integers(0, 20, GB * NS)], 1).astype(np.int32)
    lab = np.maximum(label[xy[:, 0], xy[:, 1]], 1).astype(np.int32)
    torch.manual_seed(0)
    net = Net(QCFG).to('cuda:0')
    eng = QuaTrainEngine(net, QuaScene(scenes, 'cuda:0'), GB // world, QCFG['dqtl'], lr=1e-2, process_group=pg)
    eng.load_plan(xy, lab)
    eng.run_plan(NS)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((eng.theta.cpu().numpy(), eng.loss_hist[:NS].cpu().numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_stage2_two_rank_dp_equals_single_rank_global_batch():
    (two, l2), (one, l1) = _run_ranks(_train_stage2, 2, ()), _run_ranks(_train_stage2, 1, ())
    err = np.abs(two - one).max()
    print('stage 2: 2-rank vs 1-rank parameters after 4 steps: max abs diff %.2e; losses %s vs %s' % (err, l2, l1))
    assert np.abs(l2 - l1).max() < 1e-5
    assert err < 2e-5
