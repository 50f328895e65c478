"""Forward and training-step throughput of the attention network (BASELINE configs[2]: 200-band HSI + 1-band LiDAR, cross-modal
attention on, bf16 MFMA, batch 1024) — used under rocprofv3 for the MFMA counters in profiles/.

    python tools/attn_bench.py [B] [iters] [train]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT]
from dmf import lib, synth
from dmf.engine import Scene
from function.function import data_padding, data_padding_aux
from model.gmfnet import Net


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cfg = {'patch_size': 11, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [145, 145, 200]}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': 40, 'attention': 1}, 'trans': {'embed_dim': 96, 'num_head': 3}}
    primary, aux, label = synth.make_scene(145, 145, 200, 1, 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    torch.manual_seed(0)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    rng = np.random.default_rng(0)
    xy = torch.from_numpy(np.stack([rng.integers(0, 145, B), rng.integers(0, 145, B)], 1).astype(np.int32)).cuda()
    logits = torch.empty(B, 17, device='cuda')
    pred = torch.empty(B, dtype=torch.int32, device='cuda')
    ws = torch.empty(lib.attn_workspace_bytes(net.shape, B), dtype=torch.uint8, device='cuda')
    inp = lib.input_gather(net.shape, scene.A, scene.B, xy)
    theta = net.flat_parameters()
    for _ in range(10):
        lib.forward_attn(net.shape, inp, theta, net.pool_w, ws, logits, pred)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        lib.forward_attn(net.shape, inp, theta, net.pool_w, ws, logits, pred)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # matrix-core FLOPs per patch: 3 projections (128x64x96), QK^T and PV (3 heads x 128x128x32 each), out-proj (128x96x48)
    flops = 2.0 * (3 * 128 * 64 * 96 + 2 * 3 * 128 * 128 * 32 + 128 * 96 * 48)
    print('attention forward: B=%d  %.1f us / batch  %.2f M patches/s  %.1f TFLOP/s on the matrix cores (padded shapes)'
          % (B, dt * 1e6, B / dt / 1e6, B * flops / dt / 1e12))


    if len(sys.argv) > 3 and sys.argv[3] == 'train':
        from dmf.engine import TrainEngine
        lab = torch.from_numpy(rng.integers(0, 17, B * 8).astype(np.int32)).cuda()
        xy8 = torch.from_numpy(np.stack([rng.integers(0, 145, B * 8), rng.integers(0, 145, B * 8)], 1).astype(np.int32))
        eng = TrainEngine(net, scene, B, lr=1e-3)
        eng.load_plan(xy8, lab)
        eng.run_plan(8, 0)
        torch.cuda.synchronize()
        n = 0
        t0 = time.perf_counter()
        while n < iters:
            eng.dev_cursor.zero_()
            eng.run_plan(8, 0)
            n += 8
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        # backward matrix-core work per patch on top of a forward recompute: dQs, dK (3 heads x 2 x 128x128x32, hi+lo),
        # dTa/dTb (2 x 128x32x48 per head, hi+lo), dWq/dWk (2 x 32x48x128 per head, hi+lo)
        bflops = 2.0 * (2 * (2 * 3 * 128 * 128 * 32) + 2 * (2 * 3 * 128 * 32 * 48) + 2 * (2 * 3 * 32 * 48 * 128)) \
            + 2.0 * 3 * 128 * 128 * 32     # S for own keys
        print('attention train step (tokens + attention fwd/bwd + conv backward + reduce/ADAM): B=%d  %.1f us / step  '
              '%.3f M patches/s  %.1f TFLOP/s on the matrix cores' % (B, dt * 1e6, B / dt / 1e6, B * (2 * flops + bflops) / dt / 1e12))


if __name__ == '__main__':
    main()
