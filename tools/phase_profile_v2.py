"""Diagnostic: clock stamps of EVERY wavefront of the v2 patch kernel (stamps build, never the shipped library).

    python tools/phase_profile_v2.py [B] [panms]

Prints, per wavefront, the median over workgroups of (stamp - kernel entry of that wave) in shader cycles.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('DMF_LIB', os.path.join(ROOT, 'dual-modal-fusion_amd', 'dmf', 'libdmf_hip_stamps.so'))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT]
from dmf import lib, synth
from dmf.engine import Scene
from function.function import data_padding, data_padding_aux
from model.gmfnet import Net

CONV = ['entry', 'prologue', 'gather issued', 'aux done', 'barrierW', 'barrier1', 'spec dW', 'sums', 'barrier2', 'slab adds', 'end', 'barrier0']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    panms = len(sys.argv) > 2 and sys.argv[2] in ('panms', 'qua')   # the reference's own data shape: 4-band MS + PAN at 4x, 16x16 patches
    qua = len(sys.argv) > 2 and sys.argv[2] == 'qua'            # ... or the stage-2 stream shape: 4 bands, aux at the same resolution
    cfg = {'patch_size': 11, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [145, 145, 200]}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}
    if panms:
        cfg.update({'patch_size': 16, 'Categories_Number': 12, 'DATA_DICT': {'s': {'size': [145, 145, 4]}}, 'scale': 1 if qua else 4})
    primary, aux, label = synth.make_scene(145, 145, 4 if panms else 200, 1, (1 if qua else 4) if panms else 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    rng = np.random.default_rng(0)
    xy = torch.from_numpy(np.stack([rng.integers(0, 145, B), rng.integers(0, 145, B)], 1).astype(np.int32)).cuda()
    lab = torch.from_numpy(rng.integers(1, 12 if panms else 17, B).astype(np.int32)).cuda()
    nblk = min(B, 256)
    stamps = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device='cuda')
    fn = lib._lib.dmf_debug_set_v2_stamps
    fn.restype, fn.argtypes = C.c_int32, [C.c_void_p]
    lib.check(fn(C.c_void_p(stamps.data_ptr())))
    # steady-state conditions: the kernel alternates with the reduce + ADAM launch inside a replayed hipGraph, exactly as in
    # bench.py (theta freshly rewritten, coordinates from the refilled window); the stamps of the LAST launch are kept
    from dmf.engine import TrainEngine
    n_steps = 120
    rng2 = np.random.default_rng(1)
    xy_all = np.stack([rng2.integers(0, 145, n_steps * B), rng2.integers(0, 145, n_steps * B)], 1).astype(np.int32)
    lab_all = rng2.integers(1, 12 if panms else 17, n_steps * B).astype(np.int32)
    eng = TrainEngine(net, scene, B, lr=1e-3)
    eng.load_plan(xy_all, lab_all)
    eng.run_plan(n_steps, 40)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nblk, 16, 16).astype(np.float64)
    t0 = s[:, :, 0].copy()
    t0[t0 == 0] = np.nan
    first = np.nanmin(t0, axis=1)                      # first wave entry per workgroup
    print('cycles since the workgroup\'s first wave entered the kernel (median over %d workgroups)' % nblk)
    print('wave ' + ' '.join('%13s' % n for n in CONV))
    for w in range(16):
        if not np.isfinite(t0[:, w]).any():
            continue
        row = []
        for i in range(12):
            v = s[:, w, i]
            ok = v > 0
            row.append(np.median(v[ok] - first[ok]) if ok.any() else float('nan'))
        print('%4d ' % w + ' '.join('%13.0f' % v for v in row))
    ends = np.nanmax(np.where(s[:, :, 10] > 0, s[:, :, 10], np.nan), axis=1)
    print('workgroup span (first entry -> last wave end): median %.0f, max %.0f' % (np.median(ends - first), np.max(ends - first)))
    rt = s[:, :, 13] - s[:, :, 12]
    cy = s[:, :, 10] - s[:, :, 0]
    ok = (rt > 0) & (cy > 0)
    r0 = s[:, :, 12]; r1 = s[:, :, 13]
    v0 = r0[r0 > 0]; v1 = r1[r1 > 0]
    print('whole grid, s_memrealtime (100 MHz): first wave entry -> last wave end %.2f us; workgroup entry spread %.2f us; per-workgroup span median %.2f us' % ((v1.max() - v0.min()) / 100.0, (np.where(r0 > 0, r0, np.inf).min(axis=1).max() - v0.min()) / 100.0, np.median(np.where(r1 > 0, r1, 0).max(axis=1) - np.where(r0 > 0, r0, np.inf).min(axis=1)) / 100.0))
    print('in-kernel clock: %.0f MHz (cycles / s_memrealtime ticks x 100 MHz, median over waves)' % (np.median(cy[ok] / rt[ok]) * 100))


if __name__ == '__main__':
    main()
