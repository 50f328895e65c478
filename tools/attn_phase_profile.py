"""Diagnostic: clock stamps of wave 0 along one patch of the attention training kernel (stamps build only).

    DMF_LIB=dual-modal-fusion_amd/dmf/libdmf_hip_stamps.so python tools/attn_phase_profile.py [B]
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

P1 = ['stage weights', 'projections', 'S^T + softmax', 'P V, O Wo, obar']
P2 = ['stage weights (+ wait for the previous head)', 'projections', 'a_j, g, S^T + softmax', 'abar, dS, dQs, dTa, store dq', '(barrier 1)',
      'dWq, S own keys, c, dK, dTb', '(barrier 2) store dK, bbar, c x g', '(barrier 3) dWk, dWv']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cfg = {'patch_size': 11, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [145, 145, 200]}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': 40, 'attention': 1}, 'trans': {'embed_dim': 96, 'num_head': 3}}
    primary, aux, label = synth.make_scene(145, 145, 200, 1, 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    rng = np.random.default_rng(0)
    xy = torch.from_numpy(np.stack([rng.integers(0, 145, B), rng.integers(0, 145, B)], 1).astype(np.int32)).cuda()
    lab = torch.from_numpy(rng.integers(1, 17, B).astype(np.int32)).cuda()
    nblk = min(B, 256)
    stamps = torch.zeros(nblk * 64, dtype=torch.int64, device='cuda')
    fn = lib._lib.dmf_debug_set_attn_stamps
    fn.restype, fn.argtypes = C.c_int32, [C.c_void_p]
    lib.check(fn(C.c_void_p(stamps.data_ptr())))
    logits = torch.empty(B, 17, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = torch.empty(lib.workspace_bytes(net.shape, B) // 4, device='cuda')
    aws = torch.empty(lib.attn_train_workspace_bytes(net.shape, B), dtype=torch.uint8, device='cuda')
    inp = lib.input_gather(net.shape, scene.A, scene.B, xy)
    theta = net.flat_parameters()
    for _ in range(3):
        lib.train_attn_fwd_bwd(net.shape, inp, theta, net.pool_w, lab, None, 1.0 / B, logits, loss, ws, aws)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nblk, 64).astype(np.float64)
    d = np.median(np.diff(s[:, :41], axis=1), axis=0)
    names = ['load tokens'] + ['p1 h%d %s' % (h, n) for h in range(3) for n in P1] + ['pooled corr., head fwd+bwd, u, dWo'] + \
            ['p2 h%d %s' % (h, n) for h in range(3) for n in P2] + ['(end)']
    print('cycles of wave 0 between stamps (median over %d workgroups), last patch of each workgroup' % nblk)
    for n, c in zip(names, d):
        print('  %-48s %8.0f' % (n, c))
    print('  %-48s %8.0f' % ('total', d.sum()))


if __name__ == '__main__':
    main()
