"""ORACLE — test infrastructure only (never imported by the product path).

fp32 CPU restatement of the GMFNet arithmetic, in plain PyTorch ops.

Why this exists: the reference loads its network by name
(`/root/reference/solver/mainsolver.py:31-34`, `importlib.import_module('model.' + net_name).Net(args=cfg)`)
but ships NO `model/` package.  The network arithmetic is therefore authored by this
project (DESIGN.md §2 "GMFNet") and this file is its CPU statement.  PARITY UNPINNED by the
reference for the network arithmetic itself: the reference pins only the contract around it
(`Net(args=cfg)`, `forward(ms, pan) -> logits[B, Categories_Number]`, `.parameters()`, `.state_dict()`;
`mainsolver.py:31-35,45,52,80,96`), which tests/test_golden_trajectory.py checks by driving the real
reference `Solver.train/test` with this class plugged in as `model.gmfnet.Net`.

The HIP path (dual-modal-fusion_amd/csrc) must match this file to <=1e-5 on fp32 logits.

Architecture (all fp32, ReLU, no normalisation layers):
  a: [B, C, P, P]        primary modality (HSI / MS), band-major patch   (dataset.py:175-179)
  b: [B, C2, S*P, S*P]   auxiliary modality (SAR / PAN / LiDAR)          (dataset.py:176,180)

  branch A  spec_a : grouped 1x1 conv  C -> F, G groups          + ReLU   ("spectral")
            spat_a : depthwise 3x3 conv F -> F, zero pad 1        + ReLU   ("spatial")
  branch B  lift_b : SxS stride-S conv C2 -> F                    + ReLU   (resolution lift)
            spat_b : depthwise 3x3 conv F -> F, zero pad 1        + ReLU
  single-input form (gmf.single_input = 1, the stage-2 net of tostagesolver.py:274 called as net(x)):
            a = x, b = mean over the bands of x (C2 = 1, S = 1)
  [optional cross-modal attention: tokens = P*P pixels, E = heads*32;
            Ta' = Ta + (softmax(Q K^T / sqrt(dh)) V) Wo^T,  Q = Ta Wq^T, K = Tb Wk^T, V = Tb Wv^T]
  pooling   z = [sum_pix w[pix] * Ya[:, pix] ; sum_pix w[pix] * Yb[:, pix]]      (fixed anchor-Gaussian w)
  head      h = ReLU(fc1 z) ; logits = fc2 h
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F_


def arch_from_cfg(cfg):
    """Derive architecture hyper-parameters from a reference-style cfg dict.

    Keys consumed: patch_size, Categories_Number, DATA_DICT[data_city].size[2] (reference keys,
    config.yml:27-28,77-80), plus the build's own optional keys `scale`, `aux_bands`, `gmf`, `trans`.
    """
    gmf = dict(cfg.get('gmf') or {})
    trans = dict(cfg.get('trans') or {})
    C = int(cfg['DATA_DICT'][cfg['data_city']]['size'][2])
    width = int(gmf.get('width', 40))
    groups = gmf.get('groups', 'auto')
    if groups in ('auto', None, 0):
        groups = auto_groups(C, width)
    heads = int(trans.get('num_head', 3))
    embed = int(trans.get('embed_dim', 96))
    single = int(gmf.get('single_input', 0))     # stage-2 net of the two-stage path: one input, aux = its band mean
    return dict(
        single_input=single,
        C=C, C2=1 if single else int(cfg.get('aux_bands', 1)), P=int(cfg['patch_size']),
        S=1 if single else int(cfg.get('scale', 4)),
        K=int(cfg['Categories_Number']), F=width, G=int(groups), H=int(gmf.get('hidden', 64)),
        sigma=float(gmf.get('pool_sigma', 2.5)), attention=int(gmf.get('attention', 0)),
        heads=heads, E=embed, mfma_bf16=int(gmf.get('mfma_bf16', 1)),
        half=int(gmf.get('half', 0)),
    )


def auto_groups(C, width):
    """Largest G <= 16 with C % G == 0, width % G == 0, whole 4-channel blocks per group ((width/G) % 4 == 0) and band groups
    the kernel reads as whole 16-byte chunks ((C/G) % 4 == 0)."""
    best = 1
    for g in range(1, 17):
        if C % g == 0 and width % g == 0 and (width // g) % 4 == 0 and (C // g) % 4 == 0:
            best = g
    return best


def anchor_pool_weights(P, sigma):
    """Fixed pooling profile.  The reference anchors a pixel's patch at its TOP-LEFT corner
    (function.py:103-110 pads bottom/right only; dataset.py:175 slices [x:x+p, y:y+p]), so the
    labelled pixel is patch element (0, 0): w[r, c] ~ exp(-(r^2+c^2)/(2 sigma^2)), sum = 1.
    sigma <= 0 selects a uniform mean."""
    r = torch.arange(P, dtype=torch.float64)
    if sigma > 0:
        w = torch.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / (2.0 * sigma * sigma))
    else:
        w = torch.ones(P, P, dtype=torch.float64)
    w = w / w.sum()
    return w.reshape(-1).to(torch.float32)


def band_mean(a):
    """Auxiliary input of the single-input net: per-pixel mean over bands of a [B, C, P, P] batch, summed in band
    order ((x0 + x1) + x2) + ... and divided by C (the intensity component `I` of image_convert/IHS.py:46)."""
    s = a[:, 0]
    for c in range(1, a.shape[1]):
        s = s + a[:, c]
    return (s / float(a.shape[1])).unsqueeze(1)


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


class Net(nn.Module):
    def __init__(self, args):
        super().__init__()
        a = arch_from_cfg(args)
        self.arch = a
        C, C2, P, S, K, Fw, G, H = a['C'], a['C2'], a['P'], a['S'], a['K'], a['F'], a['G'], a['H']
        self.spec_a = nn.Conv2d(C, Fw, 1, groups=G)
        self.spat_a = nn.Conv2d(Fw, Fw, 3, padding=1, groups=Fw)
        self.lift_b = nn.Conv2d(C2, Fw, S, stride=S)
        self.spat_b = nn.Conv2d(Fw, Fw, 3, padding=1, groups=Fw)
        if a['attention']:
            E = a['E']
            self.attn_wq = nn.Parameter(torch.empty(E, Fw))
            self.attn_wk = nn.Parameter(torch.empty(E, Fw))
            self.attn_wv = nn.Parameter(torch.empty(E, Fw))
            self.attn_wo = nn.Parameter(torch.empty(Fw, E))
            for w in (self.attn_wq, self.attn_wk, self.attn_wv, self.attn_wo):
                bound = 1.0 / math.sqrt(w.shape[1])
                nn.init.uniform_(w, -bound, bound)
        self.fc1 = nn.Linear(2 * Fw, H)
        self.fc2 = nn.Linear(H, K)
        self.register_buffer('pool_w', anchor_pool_weights(P, a['sigma']))

    # -- pieces, exposed so tests can probe intermediate tensors ------------------------------
    def branches(self, a, b):
        if self.arch.get('half'):
            # gmf.half: the primary modality is STORED as fp16 and spec_a runs on fp16 operands — input and weight rounded to
            # nearest even, exact products, fp32 accumulation (what v_dot2_f32_f16 / v_fma_mix_f32 do in the HIP kernel).
            # The roundings have no gradient of their own (straight through): the fp32 master weight gets the gradient of
            # its rounded copy, as under torch.autocast (tostagesolver.py:86-178 runs stage 1 that way).
            y1 = F_.conv2d(_ste_f16(a), _ste_f16(self.spec_a.weight), self.spec_a.bias, groups=self.spec_a.groups)
            ya = F_.relu(self.spat_a(F_.relu(y1)))
        else:
            ya = F_.relu(self.spat_a(F_.relu(self.spec_a(a))))
        yb = F_.relu(self.spat_b(F_.relu(self.lift_b(b))))
        return ya, yb

    def attention(self, ya, yb):
        """Cross-modal attention, Ta' = Ta + bf16(O) bf16(Wo)^T with O = softmax(Qs K^T) V per head.

        With `mfma_bf16` every matrix-core operand is rounded to bf16 (round-to-nearest-even) and every product
        accumulates in fp32, which is what the HIP kernel's `v_mfma_f32_16x16x32_bf16` does:
          q = bf16(Ta) bf16(Wq)^T, k = bf16(Tb) bf16(Wk)^T, v = bf16(Tb) bf16(Wv)^T      (projections)
          s = bf16(q / sqrt(dh)) bf16(k)^T;  p = softmax(s) in fp32;  o = bf16(p) bf16(v)
        Gradients pass straight through the roundings (the backward is not built in HIP yet)."""
        A = self.arch
        B, Fw, P, _ = ya.shape
        T = P * P
        nh, E = A['heads'], A['E']
        dh = E // nh
        r = _ste_bf16 if A['mfma_bf16'] else (lambda x: x)
        ta = ya.reshape(B, Fw, T).transpose(1, 2)            # [B, T, F]
        tb = yb.reshape(B, Fw, T).transpose(1, 2)
        q = (r(ta) @ r(self.attn_wq).t()).reshape(B, T, nh, dh).transpose(1, 2)   # [B, nh, T, dh]
        k = (r(tb) @ r(self.attn_wk).t()).reshape(B, T, nh, dh).transpose(1, 2)
        v = (r(tb) @ r(self.attn_wv).t()).reshape(B, T, nh, dh).transpose(1, 2)
        scale = torch.tensor(1.0 / math.sqrt(dh), dtype=torch.float32)
        s = r(q * scale) @ r(k).transpose(-1, -2)
        p = torch.softmax(s, dim=-1)
        o = r(p) @ r(v)
        o = o.transpose(1, 2).reshape(B, T, E)
        ta2 = ta + r(o) @ r(self.attn_wo).t()
        return ta2.transpose(1, 2).reshape(B, Fw, P, P)

    def pooled(self, ya, yb):
        B, Fw = ya.shape[:2]
        w = self.pool_w
        za = (ya.reshape(B, Fw, -1) * w).sum(-1)
        zb = (yb.reshape(B, Fw, -1) * w).sum(-1)
        return torch.cat([za, zb], dim=1)

    def forward(self, a, b=None):
        """`net(ms, pan)` (mainsolver.py:52) or, with gmf.single_input, `net(data)` (tostagesolver.py:274)."""
        if b is None:
            if not self.arch['single_input']:
                raise TypeError('forward(a) with one input needs cfg["gmf"]["single_input"] = 1')
            b = band_mean(a)
        ya, yb = self.branches(a, b)
        if self.arch['attention']:
            ya = self.attention(ya, yb)
        z = self.pooled(ya, yb)
        h = F_.relu(self.fc1(z))
        return self.fc2(h)


class _SteBf16(torch.autograd.Function):
    """bf16 operand rounding with straight-through gradient (the HIP backward rounds its own
    operands the same way; the rounding itself has no gradient)."""

    @staticmethod
    def forward(ctx, x):
        return bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return g


def _ste_bf16(x):
    return _SteBf16.apply(x)


class _SteF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _ste_f16(x):
    return _SteF16.apply(x)


def adam_step_ref(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as constructed by the reference (utils/utils.py:10-12):
    lr only, betas (0.9, 0.999), eps 1e-8, weight_decay 0, amsgrad False.  In-place on fp32 tensors;
    `step` is the 1-based step index.  Scalar constants are formed in Python float64 as torch's
    single-tensor path does: step_size = lr / bc1; denom = sqrt(v) / sqrt(bc2) + eps."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)
