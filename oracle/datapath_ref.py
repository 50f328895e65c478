"""ORACLE — test infrastructure only (never imported by the product path).

numpy / torch-CPU restatement of the reference's data path, metrics and losses around the network.
Each function cites the reference lines it follows.  Pinned by tests/golden/*.npz|json, which
oracle/make_goldens.py generates by importing the real reference in the build container
(tests/test_oracle_goldens.py re-checks this file against those fixtures on every run).

PARITY UNPINNED for `pad_reflect101` only: the reference calls `cv2.copyMakeBorder(..., BORDER_REFLECT_101)`
(function/function.py:103-110) and cv2 is not installed in the build image, so the border rule is
restated from OpenCV's published definition (`gfedcb|abcdefgh|gfedcba`, no edge repeat == numpy 'reflect').
"""
import math

import numpy as np
import torch
import torch.nn.functional as F_


# ---------------------------------------------------------------- normalise + pad (a3)
def to_tensor(image):
    """function/function.py:120-124 — global min-max over the WHOLE array (all bands together)."""
    max_i = np.max(image)
    min_i = np.min(image)
    return (image - min_i) / (max_i - min_i)


def pad_reflect101(array, bottom, right):
    """cv2.copyMakeBorder(array, 0, bottom, 0, right, BORDER_REFLECT_101) restated (see header)."""
    pads = [(0, bottom), (0, right)] + [(0, 0)] * (array.ndim - 2)
    return np.pad(array, pads, mode='reflect')


def data_padding(array, patch_size, scale=4):
    """function/function.py:99-117 — normalise, then pad bottom/right only by (P-1) where
    P = patch_size for a 3-D (H,W,C) array and patch_size*scale for a 2-D one (reference: scale = 4)."""
    P = patch_size if array.ndim == 3 else patch_size * scale
    return pad_reflect101(to_tensor(array), P - 1, P - 1)


# ---------------------------------------------------------------- pixel table (a4)
def split_data_old(label, size):
    """function/function.py:149-169 — row-major (x, y, label) table as three (N,1) float64 columns and
    the index lists of label == 0 / label != 0 pixels."""
    H, W = int(size[0]), int(size[1])
    xs, ys = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    lab = np.asarray(label)[:H, :W]
    the_matrix = [xs.reshape(-1, 1).astype(np.float64), ys.reshape(-1, 1).astype(np.float64),
                  lab.reshape(-1, 1).astype(np.float64)]
    flat = lab.reshape(-1)
    matrix_ = [np.nonzero(flat == 0)[0].tolist(), np.nonzero(flat != 0)[0].tolist()]
    return the_matrix, matrix_


# ---------------------------------------------------------------- patch slicing (a2)
def dataset_dual_item(MS, PAN, xyl, index, patch_size, scale=4):
    """train/dataset.py:168-185 — (ms[C,p,p] f32, pan[1,sp,sp] f32, label 0-dim f32, x:int, y:int)."""
    p = patch_size
    x = int(np.asarray(xyl[0][index]).reshape(-1)[0])
    y = int(np.asarray(xyl[1][index]).reshape(-1)[0])
    ms = MS[x:x + p, y:y + p, :].transpose((2, 0, 1))
    pan = PAN[scale * x:scale * x + scale * p, scale * y:scale * y + scale * p]
    pan = pan[None] if pan.ndim == 2 else pan.transpose((2, 0, 1))
    label = torch.Tensor(xyl[2][index]).squeeze()
    return (torch.from_numpy(np.ascontiguousarray(ms)).type(torch.FloatTensor),
            torch.from_numpy(np.ascontiguousarray(pan)).type(torch.FloatTensor), label, x, y)


# ---------------------------------------------------------------- split sizes (a5)
def split_sizes(n, train_rate, verify_rate):
    """solver/basesolver.py:87-90."""
    train_size = int(train_rate * n)
    valid_size = int(verify_rate * n)
    return train_size, n - train_size - valid_size, valid_size


# ---------------------------------------------------------------- metrics (a10)
def kappa(matrix):
    """indicators/kappa.py:10-22."""
    m = np.asarray(matrix, dtype=np.float64)
    n = m.sum()
    po = np.trace(m) / n
    pe = float((m.sum(axis=1) * m.sum(axis=0)).sum()) / (n * n)
    return (po - pe) / (1 - pe)


def aa_oa(matrix):
    """indicators/kappa.py:69-84 — matrix[pred][target]; per-class accuracy m[i][i]/colsum[i] for i >= 1;
    AA = mean of those; OA = sum_{i>=1} m[i][i] / sum(all)  (class 0 stays in the denominator)."""
    m = np.asarray(matrix, dtype=np.float64)
    b = m.sum(axis=0)
    acc = [m[i][i] / b[i] for i in range(1, m.shape[0])]
    c = sum(m[i][i] for i in range(1, m.shape[0]))
    return [float(np.mean(acc)), float(c / b.sum()), float(kappa(m)),
            [[b[i], m[i][i], m[i][i] / b[i]] for i in range(1, m.shape[0])]]


def confusion(pred, target, K):
    """solver/mainsolver.py:139-141 — test_matrix[pred][target] += 1 (rows = prediction)."""
    m = np.zeros([K, K])
    for p, t in zip(np.asarray(pred).reshape(-1), np.asarray(target).reshape(-1)):
        m[int(p)][int(t)] += 1
    return m


# ---------------------------------------------------------------- IHS helpers (a11, a12)
def unsampling(im, scale):
    """image_convert/IHS.py:6-12 — scale x scale mean pool (output size floor(H/scale))."""
    H, W = im.shape
    h, w = int(H / scale), int(W / scale)
    out = np.zeros([h, w])
    for i in range(0, H, scale):
        for j in range(0, W, scale):
            if int(i / scale) < h and int(j / scale) < w:
                out[int(i / scale), int(j / scale)] = np.mean(im[i:i + scale, j:j + scale], axis=(0, 1))
    return out


def pan2ms(pan, size):
    """image_convert/IHS.py:14-19 — 2x2 mean pool, then 2x2 polyphase split: band i = p[i%2::2, i//2::2]."""
    p = unsampling(pan, 2)
    out = np.zeros(size)
    for i in range(size[2]):
        out[:, :, i] = p[i % 2::2, int(i / 2)::2]
    return out


# ---------------------------------------------------------------- losses (a7, a13)
def cross_entropy(logits, target):
    """utils/utils.py:28-29 `nn.CrossEntropyLoss()` (mean) with `target.long()` (mainsolver.py:53)."""
    return F_.cross_entropy(logits, target.long())


def qua_loss(out, bs, t, alpha, beta, gamma, epsilon, tao):
    """train/loss_function.py:15-76, restated as one function.  `out` = [4*bs, K] logits of the four
    streams stacked on the batch axis; `t` = [bs] float class ids."""
    data = out.softmax(dim=-1)
    p, q, r, s = data[:bs], data[bs:2 * bs], data[2 * bs:3 * bs], data[3 * bs:]

    def kl(log_in, tgt):  # F.kl_div(input=log-prob, target=prob, 'batchmean')
        return F_.kl_div(log_in, tgt, reduction='batchmean')

    l1 = l2 = 0
    if alpha != 0:
        KL_M_P = kl((q + epsilon).log(), p)
        KL_M_GM = kl((r + epsilon).log(), p)
        KL_M_GP = kl((s + epsilon).log(), p)
        KL_P_M = kl((p + epsilon).log(), q)
        KL_P_GP = kl((r + epsilon).log(), q)
        KL_P_GM = kl((s + epsilon).log(), q)
        l1 = KL_M_P + KL_M_GM + torch.abs(KL_M_GP - KL_M_GM + tao)
        l2 = KL_P_M + KL_P_GP + torch.abs(KL_P_GM - KL_P_GP + tao)
    l3 = 0
    if beta != 0:
        KL_M_GP = kl((s + epsilon).log(), p)
        KL_P_GM = kl((s + epsilon).log(), q)
        l3 = torch.mean(torch.exp(-torch.abs(KL_M_GP / p)) + torch.exp(-torch.abs(KL_P_GM / q)))
    label = torch.zeros(p.shape)
    for i in range(label.shape[0]):
        label[i][int(t[i])] = 1
    l = label.softmax(dim=-1)       # softmax OF the one-hot (loss_function.py:52), not the one-hot
    l4 = kl((p + q).softmax(dim=-1).log(), l)
    return alpha * (l1 + l2) + beta * l3 + gamma * l4


def exponential_lr(lr0, gamma, epoch):
    """utils/utils.py:65-66 ExponentialLR(gamma=0.98): lr after `epoch` scheduler steps."""
    return lr0 * math.pow(gamma, epoch)
