"""function.function — host-side scene preparation (mirror of the reference module of the same name).

Same names, arguments and results as the reference for the functions on the hot path's input side:
`to_tensor` (function/function.py:120-124), `data_padding` (:99-117), `split_data_old` (:149-169),
`read_tif` (:34-43), `data_show` (:127-133).  Differences, all deliberate:
  * `split_data_old` is vectorised (the reference walks H x W in a Python double loop) — same tables;
  * `data_padding` takes the aux/primary resolution ratio from cfg['scale'] (the reference hard-codes 4)
    and restates cv2.BORDER_REFLECT_101 with numpy 'reflect' (same rule: no edge repeat);
  * `read_tif` reads `<name>.npy` beside the TIFF name when no TIFF reader is importable (libtiff is not
    available in this image); real TIFF ingest is listed as "next" in DESIGN.md.
"""
import os

import numpy as np


def to_tensor(image):
    max_i = np.max(image)
    min_i = np.min(image)
    return (image - min_i) / (max_i - min_i)


def data_padding(array, cfg, mode):
    """Normalise, then pad bottom/right only by (patch-1) for a 3-D array, (scale*patch-1) for a 2-D one.
    `mode` is accepted for signature compatibility (the reference ignores it too)."""
    scale = int(cfg.get('scale', 4))
    axis = len(array.shape)
    patch = cfg['patch_size'] if axis == 3 else cfg['patch_size'] * scale
    array = to_tensor(array)
    pads = [(0, patch - 1), (0, patch - 1)] + [(0, 0)] * (axis - 2)
    return np.pad(array, pads, mode='reflect')


def data_padding_aux(array, cfg):
    """Aux scene of any rank ([SH, SW] or [SH, SW, C2]) padded for scale*patch windows."""
    scale = int(cfg.get('scale', 4))
    patch = cfg['patch_size'] * scale
    array = to_tensor(array)
    pads = [(0, patch - 1), (0, patch - 1)] + [(0, 0)] * (array.ndim - 2)
    return np.pad(array, pads, mode='reflect')


def split_data_old(label, cfg):
    size = cfg['DATA_DICT'][cfg['data_city']]['size']
    H, W = int(size[0]), int(size[1])
    lab = np.asarray(label)[:H, :W]
    xs, ys = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    the_matrix = [xs.reshape(-1, 1).astype(np.float64), ys.reshape(-1, 1).astype(np.float64),
                  lab.reshape(-1, 1).astype(np.float64)]
    flat = lab.reshape(-1)
    matrix_ = [np.nonzero(flat == 0)[0].tolist(), np.nonzero(flat != 0)[0].tolist()]
    for i in range(2):
        print("label set {} size {}".format(i, len(matrix_[i])))
    return the_matrix, matrix_


def read_tif(cfg, mode):
    if mode == 'ms':
        filename = cfg['data_address'] + 'ms4.tif'
    elif mode == 'pan':
        filename = cfg['data_address'] + 'pan.tif'
    else:
        raise ValueError("mode")
    if os.path.exists(filename + '.npy'):
        return np.load(filename + '.npy')
    try:
        from libtiff import TIFF
    except ImportError as e:
        raise FileNotFoundError('%s.npy not found and no TIFF reader (libtiff) is importable' % filename) from e
    return TIFF.open(filename, mode='r').read_image()


def data_show(matrix):
    label_element, element_count = np.unique(matrix, return_counts=True)
    print("labels {} counts {} shape {} classes {}".format(label_element, element_count, np.shape(matrix),
                                                          len(label_element) - 1))
