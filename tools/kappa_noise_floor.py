"""How far apart do two runs of the CPU ORACLE ITSELF end up?  (context for bench.py's |delta kappa| between GPU and oracle)

The oracle (oracle/solver_ref.py, torch-CPU fp32) trains the BASELINE configs[1] problem of bench.py for the same 2,200 steps
from the same initial weights on the same batches, once per thread count given: the thread count changes how torch partitions
its reductions (convolution weight gradients, sums over the batch), i.e. only the floating-point summation order.  Kappa of
every run on the same 2,048 held-out patches, and the spread between the runs, are printed as one JSON line.  CPU only.

    python tools/kappa_noise_floor.py [--steps 2200] [--threads 8 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT]
import bench                                                                     # noqa: E402  (problem construction only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2200)
    ap.add_argument('--threads', type=int, nargs='+', default=[8, 3])
    ap.add_argument('--test-patches', type=int, default=2048)
    a = ap.parse_args()
    args = argparse.Namespace(**{k.replace('-', '_'): v for k, v in bench.CONFIGS['1'].items() if k != 'name'})
    args.half, args.train_rate = 0, 0.10
    cfg = bench.make_cfg(args)
    MS, PAN, xy_tab, lab_tab, train, test = bench.build_problem(args, cfg)
    B, P, S = args.batch, args.patch, args.scale
    plan = bench.make_plan(train, a.steps, B, seed=1)
    from oracle.gmfnet_ref import Net as RefNet
    from oracle import solver_ref
    from indicators.kappa import aa_oa_quiet
    torch.manual_seed(3407)
    init = {k: v.detach().clone() for k, v in RefNet(cfg).state_dict().items()}
    n_kt = min(len(test), a.test_patches)
    runs, mats, losses = [], [], []
    for n_thr in a.threads:
        torch.set_num_threads(n_thr)
        ref = RefNet(cfg)
        ref.load_state_dict(init)
        opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
        t0 = time.perf_counter()
        ls = []
        for c0 in range(0, a.steps, 100):
            n_c = min(100, a.steps - c0)
            sl = plan[c0 * B:(c0 + n_c) * B]
            l, opt = solver_ref.train_steps(ref, MS, PAN, xy_tab[sl], lab_tab[sl], B, P, S, optimizer=opt)
            ls += l
            print('threads %d: step %d, %.0f s' % (n_thr, c0 + n_c, time.perf_counter() - t0), file=sys.stderr, flush=True)
        m, _ = solver_ref.evaluate(ref, MS, PAN, xy_tab[test[:n_kt]], lab_tab[test[:n_kt]], args.classes + 1, P, S)
        runs.append({'threads': n_thr, 'kappa': aa_oa_quiet(m)[2], 'seconds': time.perf_counter() - t0})
        mats.append(m)
        losses.append(np.array(ls))
    ks = [r['kappa'] for r in runs]
    out = {'steps': a.steps, 'batch': B, 'test_patches': int(n_kt), 'runs': runs, 'max_abs_delta_kappa': float(max(ks) - min(ks)),
           'confusion_entries_differing_first_two': int(np.abs(mats[0] - mats[1]).sum() // 2) if len(mats) > 1 else 0,
           'max_abs_loss_diff_first_two': float(np.abs(losses[0] - losses[1]).max()) if len(losses) > 1 else 0.0}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
