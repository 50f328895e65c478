"""Diagnostic: clock stamps of every wave of the gradient-reduce launch (stamps build, never the shipped library), taken
from the LAST step of a replayed hipGraph at the default bench shape (steady state: alternating with the patch kernel).

    python tools/reduce_phase_profile.py [B]
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
from dmf.engine import Scene, TrainEngine
from function.function import data_padding, data_padding_aux
from model.gmfnet import Net

NAMES = ['entry', 'role known', '-', 'partials written', 'behind barrier', 'stores issued', 'end (vmcnt 0)']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cfg = {'patch_size': 11, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [145, 145, 200]}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}
    primary, aux, label = synth.make_scene(145, 145, 200, 1, 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    NBLK = 512
    stamps = torch.zeros(NBLK * 5 * 8, dtype=torch.int64, device='cuda')
    fn = lib._lib.dmf_debug_set_reduce_stamps
    fn.restype, fn.argtypes = C.c_int32, [C.c_void_p]
    lib.check(fn(C.c_void_p(stamps.data_ptr())))
    n_steps = 120
    rng2 = np.random.default_rng(1)
    xy_all = np.stack([rng2.integers(0, 145, n_steps * B), rng2.integers(0, 145, n_steps * B)], 1).astype(np.int32)
    lab_all = rng2.integers(1, 17, n_steps * B).astype(np.int32)
    eng = TrainEngine(net, scene, B, lr=1e-3)
    eng.load_plan(xy_all, lab_all)
    eng.run_plan(n_steps, 40)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(NBLK, 5, 8).astype(np.float64)
    used = np.where((s[:, :, 0] > 0).any(axis=1))[0]
    print('blocks with stamps: %d (block order: fc1 tiles, fc2 tiles, conv slabs, biases; the bookkeeping block leaves none)' % len(used))
    t00 = s[used][:, :, 0]
    first_all = t00[t00 > 0].min()
    print('cycles since the FIRST wave of the grid entered the kernel; per block the median over its waves 0-3, wave 4 apart')
    print('%5s %12s %12s %16s %16s %14s %14s | wave 4: entry, barrier, end' % ('block', 'entry', 'role known', 'partials written', 'behind barrier', 'stores issued', 'end'))
    for b in used:
        row = [np.median(s[b, :4, i] - first_all) for i in (0, 1, 3, 4, 5, 6)]
        w4 = [s[b, 4, i] - first_all for i in (0, 4, 6)]
        print('%5d ' % b + ' '.join('%12.0f' % v for v in row[:2]) + ' ' + ' '.join('%16.0f' % v for v in row[2:4]) + ' ' + ' '.join('%14.0f' % v for v in row[4:])
              + ' | ' + ' '.join('%8.0f' % v for v in w4))
    ends = s[used][:, :, 6]
    print('grid span in cycles: first entry -> last end %.0f; entry spread %.0f' % (ends.max() - first_all, t00[t00 > 0].max() - first_all))
    rt = s[used][:, :, 7]
    print('s_memrealtime entry spread: %.2f us' % ((rt[rt > 0].max() - rt[rt > 0].min()) / 100.0))


if __name__ == '__main__':
    main()
