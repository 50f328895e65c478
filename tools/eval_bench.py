"""Whole-scene classification rate of the evaluation path (mainsolver.py:155-197: every pixel -> class map): EvalEngine.label_map
and .confusion over all pixels of a synthetic scene, for several evaluation chunk sizes.

    python tools/eval_bench.py [size] [bands] [aux_bands] [width]      default: 512 224 3 32 (BASELINE configs[3] scene)
    python tools/eval_bench.py qua [size]                              stage 2: pair prediction over four 4-band streams, 16x16 patches
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT]
from dmf import synth
from dmf.engine import EvalEngine, Scene
from function.function import data_padding, data_padding_aux
from model.gmfnet import Net


def main_qua():
    from dmf.engine import QuaEvalEngine, QuaScene
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = {'patch_size': 16, 'Categories_Number': 12, 'data_city': 's', 'DATA_DICT': {'s': {'size': [size, size, 4]}},
           'gmf': {'width': 40, 'single_input': 1}, 'dqtl': {'alpha': 0.1, 'beta': 0.05, 'gamma': 1.0, 'epsilon': 1e-8, 'tao': 0.1}}
    ms, pan, label = synth.make_scene(size, size, 4, 1, 1, n_classes=11, seed=0)
    g = np.random.default_rng(1)
    scenes = [data_padding(x, cfg, 'ms') for x in (ms, ms[::-1].copy(), ms + 0.1 * g.standard_normal(ms.shape), ms * 0.5)]
    torch.manual_seed(0)
    net = Net(cfg).cuda()
    scene = QuaScene(scenes, 'cuda:0')
    xx, yy = np.meshgrid(np.arange(size), np.arange(size), indexing='ij')
    xy = torch.from_numpy(np.stack([xx.reshape(-1), yy.reshape(-1)], 1).astype(np.int32))
    n = xy.shape[0]
    ref_map = None
    for B in (256, 2048, 8192):
        ev = QuaEvalEngine(net, scene, B, cfg['dqtl'])
        ev.label_map(xy[:B], size, size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lm = ev.label_map(xy, size, size)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if ref_map is None:
            ref_map = lm.clone()
        print('stage 2, chunk %5d: class map of %d pixels (2 streams each) %.2f ms (%.1f M pixels/s); map identical to chunk 256: %s'
              % (B, n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, bool(torch.equal(lm, ref_map))), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'qua':
        return main_qua()
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    C2 = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    width = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    P = 11
    cfg = {'patch_size': P, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [size, size, C]}},
           'scale': 1, 'aux_bands': C2, 'gmf': {'width': width}}
    primary, aux, label = synth.make_scene(size, size, C, C2, 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    torch.manual_seed(0)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    xx, yy = np.meshgrid(np.arange(size), np.arange(size), indexing='ij')
    xy = torch.from_numpy(np.stack([xx.reshape(-1), yy.reshape(-1)], 1).astype(np.int32)).cuda()
    lab = torch.from_numpy(np.maximum(label.reshape(-1), 1).astype(np.int32)).cuda()
    n = xy.shape[0]
    bytes_per_patch = 4.0 * (P * P * C + P * P * C2)
    ref_map = None
    for B in (256, 1024, 4096, 16384):
        ev = EvalEngine(net, scene, B)
        ev.label_map(xy[:B * 2], size, size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lm = ev.label_map(xy, size, size)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m = ev.confusion(xy, lab)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if ref_map is None:
            ref_map = lm.clone()
        same = bool(torch.equal(lm, ref_map))
        print('chunk %6d: class map of %d pixels %.2f ms (%.1f M pixels/s, %.2f TB/s of window reads = %.2f of 8 TB/s), confusion %.2f ms; '
              'map identical to chunk 256: %s' % (B, n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, n * bytes_per_patch / (t1 - t0) / 1e12,
                                                   n * bytes_per_patch / (t1 - t0) / 8e12, (t2 - t1) * 1e3, same), flush=True)


if __name__ == '__main__':
    main()
