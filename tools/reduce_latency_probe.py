"""Diagnostic (stamps build): where do the reduce launch's load cycles go?  Replays a graph of [patch, reduce] or
[patch, reduce, reduce] steps and prints, for the LAST reduce launch, per block kind the median cycles since block entry.
    python tools/reduce_latency_probe.py {1|2}          (number of reduce launches per step; DMF_REDUCE_DBG=1: wait for the ADAM state alone)
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


def main():
    n_red = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B = 256
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
    rng = np.random.default_rng(1)
    n = 40
    xy = torch.from_numpy(np.stack([rng.integers(0, 145, n * B), rng.integers(0, 145, n * B)], 1).astype(np.int32)).cuda()
    lab = torch.from_numpy(rng.integers(1, 17, n * B).astype(np.int32)).cuda()
    theta = net.flat_parameters()
    m, v, grad = torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta)
    logits = torch.empty(B, 17, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = torch.empty(lib.workspace_bytes(net.shape, B) // 4, device='cuda')
    step = torch.zeros(1, dtype=torch.int32, device='cuda')

    def one(k):
        inp = lib.input_gather(net.shape, scene.A, scene.B, xy[k * B:(k + 1) * B])
        lib.train_fwd_bwd(net.shape, inp, theta, net.pool_w, lab[k * B:(k + 1) * B], 1.0 / B, logits, loss, ws, adam_step_dev=step)
        for _ in range(n_red):
            lib.grad_reduce_adam(net.shape, B, ws, theta, m, v, None, 1e-3, 0.9, 0.999, 1e-8, 0, adam_step_dev=step)
    one(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(n):
            one(k)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(NBLK, 5, 8).astype(np.float64)
    used = np.where((s[:, :, 0] > 0).any(axis=1))[0]
    kinds = {'fc tiles': [b for b in used if b < 104], 'conv pieces': [b for b in used if b >= 104]}
    print('reduce launches per step: %d, DMF_REDUCE_DBG=%s; cycles since block entry, median over blocks (waves 0-3)' % (n_red, os.environ.get('DMF_REDUCE_DBG', '0')))
    for name, bl in kinds.items():
        rel = []
        for b in bl:
            e = s[b, :4, 0].min()
            rel.append([np.median(s[b, :4, i]) - e for i in (2, 1, 3, 4, 5, 6)])
        rel = np.array(rel)
        print('%-12s kernel arguments %6.0f | role known %6.0f | partials written %6.0f | behind barrier %6.0f | stores issued %6.0f | end %6.0f'
              % ((name,) + tuple(np.median(rel, axis=0))))


if __name__ == '__main__':
    main()
