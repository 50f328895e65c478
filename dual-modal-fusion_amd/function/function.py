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


def split_data(train_label, test_label, label, cfg):
    """function.py:172-194 of the reference (`data_new: 1`): the row-major pixel table of the whole scene and three index
    lists — pixels in neither mask, pixels of the TRAIN mask, pixels of the TEST mask only (a pixel in both counts as
    train: the reference's `if train ... elif test`)."""
    size = cfg['DATA_DICT'][cfg['data_city']]['size']
    H, W = int(size[0]), int(size[1])
    lab = np.asarray(label)[:H, :W]
    tr = np.asarray(train_label)[:H, :W].reshape(-1) != 0
    te = np.asarray(test_label)[:H, :W].reshape(-1) != 0
    xs, ys = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    the_matrix = [xs.reshape(-1, 1).astype(np.float64), ys.reshape(-1, 1).astype(np.float64),
                  lab.reshape(-1, 1).astype(np.float64)]
    matrix_ = [np.nonzero(~tr & ~te)[0].tolist(), np.nonzero(tr)[0].tolist(), np.nonzero(~tr & te)[0].tolist()]
    for i in range(3):
        print("label set {} size {}".format(i, len(matrix_[i])))
    return the_matrix, matrix_


def read_tif(cfg, mode):
    if mode == 'ms':
        filename = cfg['data_address'] + 'ms4.tif'
    elif mode == 'pan':
        filename = cfg['data_address'] + 'pan.tif'
    else:
        raise ValueError("mode")
    if os.path.exists(filename + '.npy'):                      # pre-converted scene (tools/make_synthetic_scene.py)
        return np.load(filename + '.npy')
    from function.tiffio import read_image                     # function.py:40-42 uses libtiff.TIFF.read_image
    return read_image(filename)


def label_mat2np(cfg):
    """function/function.py:11-17: `label.mat` -> `label.npy` (uint8, transposed).  The reference opens the file with
    h5py (MATLAB v7.3); h5py is not importable here, so MATLAB v5 / v7 files are read with scipy.io.loadmat — those are
    stored in MATLAB's column-major order already transposed back by scipy, hence no transpose on that route — and a
    v7.3 file fails with a clear message."""
    path = cfg['data_address']
    try:
        import scipy.io
        label = np.array(scipy.io.loadmat(path + 'label.mat')['label'], dtype='uint8')
    except NotImplementedError as e:                           # scipy: "Please use HDF reader for matlab v7.3 files"
        raise NotImplementedError(path + 'label.mat is a MATLAB v7.3 (HDF5) file and h5py is not installed; '
                                  'convert it to label.npy (np.transpose of the stored array) elsewhere') from e
    print(label.shape)
    np.save(path + 'label.npy', label)
    return label


def data_show(matrix):
    label_element, element_count = np.unique(matrix, return_counts=True)
    print("labels {} counts {} shape {} classes {}".format(label_element, element_count, np.shape(matrix),
                                                          len(label_element) - 1))
