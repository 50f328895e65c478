"""Synthetic dual-modal scenes (SURVEY.md §8d / BASELINE.md §3).

No dataset can be fetched in this environment, so every benchmark and test uses a seeded synthetic
scene with the reference's on-disk geometry (`function/function.py:34-43`: `ms4.tif` = [H, W, C],
`pan.tif` = [S*H, S*W] (reference S = 4), `label.npy` = [H, W] uint8 with 0 = unlabelled):

  label  : 5x5-pixel blocks of classes 1..n_classes, then 30 % of pixels set to 0
  primary: per-class N(0,1) prototype spectrum + 0.5*N(0,1) noise, float32, [H, W, C]
  aux    : class/16 + 0.3*N(0,1), float32, [S*H, S*W] (C2 == 1) or [S*H, S*W, C2]
"""
import os

import numpy as np


def make_scene(H=145, W=145, C=200, C2=1, S=1, n_classes=16, seed=0, block=5, unlabelled=0.30):
    rng = np.random.default_rng(seed)
    bh, bw = -(-H // block), -(-W // block)
    blocks = rng.integers(1, n_classes + 1, size=(bh, bw))
    cls = np.kron(blocks, np.ones((block, block), dtype=np.int64))[:H, :W]
    label = cls.copy()
    label[rng.random((H, W)) < unlabelled] = 0
    proto = rng.standard_normal((n_classes + 1, C)).astype(np.float32)
    primary = proto[cls] + 0.5 * rng.standard_normal((H, W, C)).astype(np.float32)
    cls_hi = np.kron(cls, np.ones((S, S), dtype=np.int64))
    shape = (S * H, S * W) if C2 == 1 else (S * H, S * W, C2)
    base = (cls_hi / 16.0).astype(np.float32)
    if C2 != 1:
        base = base[..., None] * np.linspace(1.0, 0.5, C2, dtype=np.float32)
    aux = base + 0.3 * rng.standard_normal(shape).astype(np.float32)
    return primary.astype(np.float32), aux.astype(np.float32), label.astype(np.uint8)


def class_colors(n):
    """n RGB triples incl. class 0 = black; `Categories_Number = len(color)` (utils/config.py:25)."""
    rng = np.random.default_rng(1234)
    cols = rng.integers(32, 256, size=(n, 3)).tolist()
    cols[0] = [0, 0, 0]
    return cols


def write_scene(dirname, primary, aux, label):
    """Write a scene the way the build's reader expects it: the reference's file names + '.npy'
    (`ms4.tif.npy`, `pan.tif.npy`; the reference's TIFF reader needs libtiff, which this image lacks,
    and plain `pan.npy` is taken by the two-stage path's pan2ms cache, function.py:207-212)."""
    os.makedirs(dirname, exist_ok=True)
    np.save(os.path.join(dirname, 'ms4.tif.npy'), primary)
    np.save(os.path.join(dirname, 'pan.tif.npy'), aux)
    np.save(os.path.join(dirname, 'label.npy'), label)
