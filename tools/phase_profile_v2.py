"""Diagnostic: clock stamps of EVERY wavefront of the v2 patch kernel (stamps build, never the shipped library).

    python tools/phase_profile_v2.py [B]

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

CONV = ['entry', 'offsets', 'gather issued', 'aux done', 'spec fwd', 'barrier1', 'spec dW', 'sums', 'barrier2', 'slab adds', 'end']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cfg = {'patch_size': 11, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [145, 145, 200]}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}
    primary, aux, label = synth.make_scene(145, 145, 200, 1, 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    rng = np.random.default_rng(0)
    xy = torch.from_numpy(np.stack([rng.integers(0, 145, B), rng.integers(0, 145, B)], 1).astype(np.int32)).cuda()
    lab = torch.from_numpy(rng.integers(1, 17, B).astype(np.int32)).cuda()
    nblk = min(B, 256)
    stamps = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device='cuda')
    fn = lib._lib.dmf_debug_set_v2_stamps
    fn.restype, fn.argtypes = C.c_int32, [C.c_void_p]
    lib.check(fn(C.c_void_p(stamps.data_ptr())))
    logits = torch.empty(B, 17, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = torch.empty(lib.workspace_bytes(net.shape, B) // 4, device='cuda')
    inp = lib.input_gather(net.shape, scene.A, scene.B, xy)
    theta = net.flat_parameters()
    for _ in range(5):
        lib.train_fwd_bwd(net.shape, inp, theta, net.pool_w, lab, 1.0 / B, logits, loss, ws)
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
        for i in range(11):
            v = s[:, w, i]
            ok = v > 0
            row.append(np.median(v[ok] - first[ok]) if ok.any() else float('nan'))
        print('%4d ' % w + ' '.join('%13.0f' % v for v in row))
    ends = np.nanmax(np.where(s[:, :, 10] > 0, s[:, :, 10], np.nan), axis=1)
    print('workgroup span (first entry -> last wave end): median %.0f, max %.0f' % (np.median(ends - first), np.max(ends - first)))
    print('all workgroups: first entry -> last end %.0f' % (np.nanmax(ends) - np.nanmin(first)))


if __name__ == '__main__':
    main()
