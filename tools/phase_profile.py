"""Diagnostic: per-phase cycle shares of the fused patch kernel (stamps build, never the shipped library).

    DMF_LIB=dual-modal-fusion_amd/dmf/libdmf_hip_stamps.so python tools/phase_profile.py [B] [width]
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

NAMES = ['issue gather + aux branch fwd', 'wait window (barrier)', 'spec_a + spat_a fwd + barrier', 'fc1, wave-0 fc2/CE/dh',
         'dz partials + barrier', 'dw3x3 dW/db', 'dY1 in place + bias/lift grads', '-', 'spec_a dW loop',
         'partials + slab + final barrier']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cfg = {'patch_size': 11, 'Categories_Number': 17, 'data_city': 's', 'DATA_DICT': {'s': {'size': [145, 145, 200]}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': width, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}
    primary, aux, label = synth.make_scene(145, 145, 200, 1, 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    rng = np.random.default_rng(0)
    xy = torch.from_numpy(np.stack([rng.integers(0, 145, B), rng.integers(0, 145, B)], 1).astype(np.int32)).cuda()
    lab = torch.from_numpy(rng.integers(1, 17, B).astype(np.int32)).cuda()
    nblk = min(B, 256)
    stamps = torch.zeros(nblk * 16, dtype=torch.int64, device='cuda')
    fn = lib._lib.dmf_debug_set_stamps
    fn.restype, fn.argtypes = C.c_int32, [C.c_void_p]
    lib.check(fn(C.c_void_p(stamps.data_ptr())))
    logits = torch.empty(B, 17, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = torch.empty(lib.workspace_bytes(net.shape, B) // 4, device='cuda')
    inp = lib.input_gather(net.shape, scene.A, scene.B, xy)
    theta = net.flat_parameters()
    for _ in range(5):
        lib.train_fwd_bwd(net.shape, inp, theta, net.pool_w, lab, 1.0 / B, logits, loss, ws)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nblk, 16)[:, :11].astype(np.float64)
    d = np.diff(s, axis=1)
    med = np.median(d, axis=0)
    tot = med.sum()
    print('per-phase cycles of wave 0 (median over %d workgroups; s_memtime ticks), last patch of each workgroup' % nblk)
    for n, c in zip(NAMES, med):
        print('  %-34s %9.0f  %5.1f %%' % (n, c, 100 * c / tot))
    print('  %-34s %9.0f' % ('total', tot))
    full = stamps.cpu().numpy().reshape(nblk, 16).astype(np.float64)
    print('  %-34s %9.0f' % ('prologue (kernel entry -> tables in LDS)', np.median(full[:, 12] - full[:, 11])))
    print('  %-34s %9.0f' % ('kernel entry -> end of patch', np.median(full[:, 10] - full[:, 11])))
    print('  %-34s %9.0f %9.0f %9.0f' % ('entry -> table loads issued + cursor read | -> gather issued | -> tables in LDS', np.median(full[:, 13] - full[:, 11]), np.median(full[:, 14] - full[:, 13]), np.median(full[:, 12] - full[:, 14])))
    span = (s[:, 10].max() - s[:, 0].min())
    print('first start -> last end across workgroups: %.0f ticks' % span)


if __name__ == '__main__':
    main()
