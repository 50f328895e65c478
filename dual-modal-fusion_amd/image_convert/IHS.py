"""image_convert.IHS — resolution helpers of the two-stage path (mirror of the reference module).

`unsampling` (scale x scale mean pool, IHS.py:6-12) and `pan2ms` (2x2 mean pool + 2x2 polyphase split, :14-19)
give the reference's fp64 results; the Python pixel loops are replaced by numpy block reductions on the host and
by `dmf_pan2ms` (include/dmf.h) on the GPU.  `IHS_tran` (:40-54, not called anywhere in the reference) is kept for
completeness: algebraically it returns PAN, whatever the random unpooling positions are.
"""
import numpy as np


def unsampling(im, scale):
    H, W = im.shape
    h, w = int(H / scale), int(W / scale)
    blocks = np.asarray(im, dtype=np.float64)[:h * scale, :w * scale].reshape(h, scale, w, scale)
    # running sum in the block's row-major order, as numpy.mean over a (scale, scale) slice does
    acc = np.zeros((h, w))
    for i in range(scale):
        for j in range(scale):
            acc = acc + blocks[:, i, :, j]
    return acc / (scale * scale)


def pan2ms(pan, size):
    p = unsampling(pan, 2)
    result = np.zeros(size)
    for i in range(size[2]):
        result[:, :, i] = p[i % 2::2, int(i / 2)::2]
    return result


def pan2ms_gpu(pan, size, device='cuda:0'):
    """Same result as `pan2ms` for size[2] == 4, computed by the HIP kernel (fp64)."""
    import torch
    from dmf import lib
    if size[2] != 4:
        raise ValueError('pan2ms splits into exactly 4 polyphase bands')
    pan_d = torch.from_numpy(np.ascontiguousarray(pan, dtype=np.float64)).to(device)
    out = torch.empty(size[0], size[1], 4, dtype=torch.float64, device=device)
    lib.pan2ms(pan_d, size[0], size[1], out)
    return out.cpu().numpy()


def unpooling(pic, time, rng=None):
    rng = rng or np.random.default_rng()
    H, W, C = pic.shape
    out = np.zeros([H * time, W * time, C])
    m = rng.integers(0, time, size=(C, H, W))
    n = rng.integers(0, time, size=(C, H, W))
    jj, kk = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    for i in range(C):
        out[time * jj + m[i], time * kk + n[i], i] = pic[:, :, i]
    return out


def raw_3copy(image_raw, n):
    return image_raw[:, :, np.newaxis].repeat([n], axis=2)


def IHS_tran(MS, PAN, rng=None):
    ms_up = unpooling(MS, MS.shape[2], rng)
    I = ms_up[:, :, 0]
    for i in range(1, MS.shape[2]):
        I = (I * i + ms_up[:, :, i]) / (i + 1)
    result = ms_up + raw_3copy(PAN - I, MS.shape[2])
    out = result[:, :, 0]
    for i in range(1, result.shape[2]):
        out = (out * i + result[:, :, i]) / (i + 1)
    return out
