"""Step rate of stage 2 of the two-stage path (BASELINE configs[4] shape: four 4-band streams, 16x16 patches, qua_loss).

    python tools/stage2_bench.py [bs] [steps]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT]
from dmf import synth
from dmf.engine import QuaEvalEngine, QuaScene, QuaTrainEngine
from function.function import data_padding
from image_convert.IHS import pan2ms_gpu
from model.gmfnet import Net


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    H = W = 256
    cfg = {'patch_size': 16, 'Categories_Number': 12, 'data_city': 's', 'DATA_DICT': {'s': {'size': [H, W, 4]}},
           'gmf': {'width': 40, 'single_input': 1}, 'dqtl': {'alpha': 0.1, 'beta': 0.05, 'gamma': 1.0, 'epsilon': 1e-8, 'tao': 0.1}}
    ms, pan, label = synth.make_scene(H, W, 4, 1, 4, n_classes=11, seed=0)
    t0 = time.perf_counter()
    pan4 = pan2ms_gpu(pan, [H, W, 4])
    t_p2m = time.perf_counter() - t0
    g = np.random.default_rng(1)
    scenes = [data_padding(x, cfg, 'ms') for x in (ms, pan4, ms + 0.1 * g.standard_normal(ms.shape), pan4 + 0.1 * g.standard_normal(pan4.shape))]
    torch.manual_seed(0)
    net = Net(cfg).cuda()
    scene = QuaScene(scenes, 'cuda:0')
    eng = QuaTrainEngine(net, scene, bs, cfg['dqtl'], lr=1e-3)
    xy = np.stack([g.integers(0, H, steps * bs), g.integers(0, W, steps * bs)], 1).astype(np.int32)
    lab = np.maximum(label[xy[:, 0], xy[:, 1]], 1).astype(np.int32)
    eng.load_plan(xy, lab)
    spg = 50 if eng.unit and steps % 50 == 0 else 0        # captured hipGraph of 50 steps where the unit-gradient step exists
    eng.run_plan(min(50, steps), spg)
    torch.cuda.synchronize()
    eng.load_plan(xy, lab)
    t0 = time.perf_counter()
    eng.run_plan(steps, spg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses = eng.losses().numpy()
    print('stage 2 train step (forward 4x%d patches, qua_loss, backward, reduce+ADAM): %.1f us / step, %.2f M stream-patches/s, '
          '%.3f M pixels/s; loss %.4f -> %.4f; pan2ms of a %dx%d PAN on the GPU incl. copies: %.1f ms'
          % (bs, dt * 1e6, 4 * bs / dt / 1e6, bs / dt / 1e6, losses[0], losses[-1], 4 * H, 4 * W, t_p2m * 1e3))
    ev = QuaEvalEngine(net, scene, 2048, cfg['dqtl'])
    xy_e = torch.from_numpy(xy[:2048])
    for _ in range(3):
        ev.predict(xy_e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ev.predict(xy_e)
    torch.cuda.synchronize()
    print('stage 2 prediction: %.1f us per 2048 pixels' % ((time.perf_counter() - t0) / 20 * 1e6))


if __name__ == '__main__':
    main()
