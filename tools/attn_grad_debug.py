"""Diagnostic: per-block error map of the attention training gradients against the oracle (GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), ROOT, os.path.join(ROOT, 'tests')]
import test_gpu_parity as T   # noqa: E402
from dmf import lib            # noqa: E402

name, B = sys.argv[1] if len(sys.argv) > 1 else 'tiny1', int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg, ref, hip = T._attn_nets(name)
a, b, t = T.rand_batch(name, B)
ref.zero_grad()
wl = ref(a, b)
torch.nn.functional.cross_entropy(wl, t).backward()
K = T.SHAPES[name][4]
inp_a, inp_b = a.cuda(), b.cuda()
inp = lib.input_patches(hip.shape, inp_a, inp_b)
theta = hip.flat_parameters()
logits = torch.empty(B, K, device='cuda'); lossv = torch.empty(B, device='cuda')
ws = torch.empty(lib.workspace_bytes(hip.shape, B) // 4, device='cuda')
aws = torch.empty(lib.attn_train_workspace_bytes(hip.shape, B), dtype=torch.uint8, device='cuda')
lib.train_attn_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), None, 1.0 / B, logits, lossv, ws, aws)
grad = torch.empty_like(theta)
lib.grad_reduce(hip.shape, B, ws, grad)
off = hip._offsets
rp = dict(ref.named_parameters())
for i, (k, p) in enumerate(zip(hip._order(), hip._named())):
    want = rp[k].grad
    got = grad[off[i]:off[i] + p.numel()].view(p.shape).cpu()
    err = (got - want).abs()
    print('%-14s max|want| %.3e  max err %.3e' % (k, want.abs().max(), err.max()))
    if k in ('attn_wq', 'attn_wk', 'attn_wv'):
        e = err.view(3, 2, 16, 40)   # head, d tile, d, f
        for h in range(3):
            print('   head %d: d-tile0 f<16 %.1e f16-31 %.1e f32+ %.1e | d-tile1 %.1e %.1e %.1e' % (
                h, e[h, 0, :, :16].max(), e[h, 0, :, 16:32].max(), e[h, 0, :, 32:].max(),
                e[h, 1, :, :16].max(), e[h, 1, :, 16:32].max(), e[h, 1, :, 32:].max()))
        r = (got / (want + 1e-20)).view(3, 2, 16, 40)
        print('   ratio samples head0 tile1:', r[0, 1, :3, :3].flatten().tolist())
    if k == 'attn_wo':
        e = err.view(40, 3, 32)
        print('   per head', [float(e[:, h].max()) for h in range(3)])

# ---- which product did the kernel form?  candidates from oracle intermediates (patch-summed)
import math
A_ = ref.arch
ya, yb = ref.branches(a, b)
Bn, Fw, P_, _ = ya.shape
Tt = P_ * P_
r_ = lambda x: x.to(torch.bfloat16).float()
ta = ya.reshape(Bn, Fw, Tt).transpose(1, 2).detach()
tb = yb.reshape(Bn, Fw, Tt).transpose(1, 2).detach()
q = (r_(ta) @ r_(ref.attn_wq).t()).detach().requires_grad_(True)
k_ = (r_(tb) @ r_(ref.attn_wk).t()).detach().requires_grad_(True)
got_q = grad[off[12]:off[13]].view(96, 40).cpu()
got_k = grad[off[13]:off[14]].view(96, 40).cpu()
want_q = rp['attn_wq'].grad
# dq, dk from autograd: rerun the oracle with hooks
store = {}
orig = type(ref).attention
def att(self, ya, yb):
    A = self.arch
    B, Fw, P, _ = ya.shape
    T = P * P
    nh, E = A['heads'], A['E']
    dh = E // nh
    from oracle.gmfnet_ref import _ste_bf16 as r
    ta = ya.reshape(B, Fw, T).transpose(1, 2)
    tb = yb.reshape(B, Fw, T).transpose(1, 2)
    qf = r(ta) @ r(self.attn_wq).t(); kf = r(tb) @ r(self.attn_wk).t()
    qf.retain_grad(); kf.retain_grad(); store['q'] = qf; store['k'] = kf; store['ta'] = ta; store['tb'] = tb
    q = qf.reshape(B, T, nh, dh).transpose(1, 2); k = kf.reshape(B, T, nh, dh).transpose(1, 2)
    v = (r(tb) @ r(self.attn_wv).t()).reshape(B, T, nh, dh).transpose(1, 2)
    scale = torch.tensor(1.0 / math.sqrt(dh), dtype=torch.float32)
    s = r(q * scale) @ r(k).transpose(-1, -2)
    p = torch.softmax(s, dim=-1)
    o = (r(p) @ r(v)).transpose(1, 2).reshape(B, T, E)
    return (ta + r(o) @ r(self.attn_wo).t()).transpose(1, 2).reshape(B, Fw, P, P)
type(ref).attention = att
ref.zero_grad()
torch.nn.functional.cross_entropy(ref(a, b), t).backward()
dq, dk = store['q'].grad, store['k'].grad            # [B, T, E]
ta_r, tb_r = r_(store['ta'].detach()), r_(store['tb'].detach())
cands = {
    'dq^T Ta': torch.einsum('bte,btf->ef', dq, ta_r), 'dq^T Tb': torch.einsum('bte,btf->ef', dq, tb_r),
    'dk^T Ta': torch.einsum('bte,btf->ef', dk, ta_r), 'dk^T Tb': torch.einsum('bte,btf->ef', dk, tb_r),
}
for nm, c in cands.items():
    print('%-8s vs got_q %.2e   vs got_k %.2e   (max %.2e)' % (nm, (c - got_q).abs().max(), (c - got_k).abs().max(), c.abs().max()))
print('got_q[0:4,0:6]\n', got_q[0:4, 0:6], '\nwant_q[0:4,0:6]\n', want_q[0:4, 0:6])
print('got_q[16:20,0:6]\n', got_q[16:20, 0:6], '\nwant_q[16:20,0:6]\n', want_q[16:20, 0:6])

if os.environ.get('DMF_LIB', '').endswith('_dbg.so'):
    nslab = 4 * 96 * 40
    base = ws.numel() - 256 * nslab
    dbg = ws[base + nslab: base + 2 * nslab].cpu()
    sd = dbg[:32 * 128].view(32, 128); st = dbg[32 * 128:2 * 32 * 128].view(32, 128)
    print('sDhi rows 0..3, t 0..5:\n', sd[:4, :6], '\n rows 16,17:\n', sd[16:18, :6])
    print('oracle dq^T[d][t] (patch 0, head 0):\n', dq[0, :6, :4].t(), '\n d=16,17:\n', dq[0, :6, 16:18].t())
    tile = dbg[2 * 32 * 128:2 * 32 * 128 + 6 * 256].view(6, 64, 4)
    print('wave0 tile lanes 0,1,16 regs:', tile[0, 0], tile[0, 1], tile[0, 16])
    print('sD @ Ta^T rows0..1,16 cols 0..1 :', (sd @ st.t())[[0, 1, 16]][:, :2] if st.shape[0] >= 2 else None)
