"""model.gmfnet — the plug-in the reference solver loads by name.

Boundary (reference): `importlib.import_module('model.' + net_name).Net(args=cfg)` (solver/mainsolver.py:31-34),
called as `net(ms, pan)` -> logits [B, Categories_Number] (mainsolver.py:52), with `.parameters()`, `.to()`,
`.train()/.eval()`, `.state_dict()/.load_state_dict()` (mainsolver.py:35,45,47,63,80,96).  The reference ships
no `model/` package, so the architecture is this build's; its CPU statement for tests is
oracle/gmfnet_ref.py.

All arithmetic runs in libdmf_hip.so (hand-written HIP, include/dmf.h); torch supplies parameters,
autograd plumbing and device memory only.  There is no CPU path: calling the net off-GPU raises.
"""
import math

import torch
import torch.nn as nn

from dmf import lib
from dmf.arch import anchor_pool_weights, arch_from_cfg

PARAM_ORDER = ('spec_a.weight', 'spec_a.bias', 'spat_a.weight', 'spat_a.bias', 'lift_b.weight', 'lift_b.bias',
               'spat_b.weight', 'spat_b.bias', 'fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias')
ATTN_ORDER = ('attn_wq', 'attn_wk', 'attn_wv', 'attn_wo')


class _GmfFunction(torch.autograd.Function):
    """forward: dmf_forward; backward: dmf_backward_dlogits + dmf_grad_reduce -> one flat gradient."""

    @staticmethod
    def forward(ctx, net, a, b, theta):
        inp = lib.input_patches(net.shape, a, b, half=net.arch.get('half', 0))
        logits = torch.empty(a.shape[0], net.arch['K'], device=a.device, dtype=torch.float32)
        lib.forward(net.shape, inp, theta, net.pool_w, logits)
        ctx.net = net
        ctx.save_for_backward(a, b, theta)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        net = ctx.net
        a, b, theta = ctx.saved_tensors
        inp = lib.input_patches(net.shape, a, b, half=net.arch.get('half', 0))
        B = a.shape[0]
        ws = net.workspace(B)
        lib.backward_dlogits(net.shape, inp, theta, net.pool_w, dlogits.contiguous().float(), ws)
        grad = torch.empty_like(theta)
        lib.grad_reduce(net.shape, B, ws, grad)
        return None, None, None, grad


class _GmfAttnFunction(torch.autograd.Function):
    """Attention network.  forward: dmf_forward_attn; backward: dmf_train_attn_fwd_bwd with the caller's dL/dlogits
    (it recomputes the forward inside the fused fwd+bwd kernels) + dmf_grad_reduce -> one flat gradient."""

    @staticmethod
    def forward(ctx, net, a, b, theta):
        inp = lib.input_patches(net.shape, a, b, half=net.arch.get('half', 0))
        B = a.shape[0]
        logits = torch.empty(B, net.arch['K'], device=a.device, dtype=torch.float32)
        ws = torch.empty(lib.attn_workspace_bytes(net.shape, B), device=a.device, dtype=torch.uint8)
        lib.forward_attn(net.shape, inp, theta, net.pool_w, ws, logits)
        ctx.net = net
        ctx.save_for_backward(a, b, theta)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        net = ctx.net
        a, b, theta = ctx.saved_tensors
        inp = lib.input_patches(net.shape, a, b, half=net.arch.get('half', 0))
        B = a.shape[0]
        ws = net.workspace(B)
        attn_ws = torch.empty(lib.attn_train_workspace_bytes(net.shape, B), device=a.device, dtype=torch.uint8)
        logits = torch.empty(B, net.arch['K'], device=a.device, dtype=torch.float32)
        lib.train_attn_fwd_bwd(net.shape, inp, theta, net.pool_w, None, dlogits.contiguous().float(), 1.0, logits, None,
                               ws, attn_ws)
        grad = torch.empty_like(theta)
        lib.grad_reduce(net.shape, B, ws, grad)
        return None, None, None, grad


class Net(nn.Module):
    def __init__(self, args):
        super().__init__()
        a = arch_from_cfg(args)
        self.arch = a
        self.shape = lib.make_shape(a)
        # parameter containers, created in the same order and with the same initialisers as their
        # torch namesakes (so a seeded run starts from the same weights as the CPU statement)
        self.spec_a = nn.Conv2d(a['C'], a['F'], 1, groups=a['G'])
        self.spat_a = nn.Conv2d(a['F'], a['F'], 3, padding=1, groups=a['F'])
        self.lift_b = nn.Conv2d(a['C2'], a['F'], a['S'], stride=a['S'])
        self.spat_b = nn.Conv2d(a['F'], a['F'], 3, padding=1, groups=a['F'])
        if a['attention']:
            # created here, in the order and with the initialiser of the CPU statement (oracle/gmfnet_ref.py)
            E = a['E']
            self.attn_wq = nn.Parameter(torch.empty(E, a['F']))
            self.attn_wk = nn.Parameter(torch.empty(E, a['F']))
            self.attn_wv = nn.Parameter(torch.empty(E, a['F']))
            self.attn_wo = nn.Parameter(torch.empty(a['F'], E))
            for w in (self.attn_wq, self.attn_wk, self.attn_wv, self.attn_wo):
                nn.init.uniform_(w, -1.0 / math.sqrt(w.shape[1]), 1.0 / math.sqrt(w.shape[1]))
        self.fc1 = nn.Linear(2 * a['F'], a['H'])
        self.fc2 = nn.Linear(a['H'], a['K'])
        self.register_buffer('pool_w', anchor_pool_weights(a['P'], a['sigma']))
        self._offsets = lib.param_layout(self.shape)
        self._flat = None
        self._ws = {}
        n = sum(p.numel() for p in self.parameters())
        if n != self._offsets[16]:
            raise lib.DmfError('parameter layout mismatch: torch %d vs library %d' % (n, self._offsets[16]))

    # ---- flat parameter vector ------------------------------------------------------------------
    def _order(self):
        return PARAM_ORDER + (ATTN_ORDER if self.arch['attention'] else ())

    def _named(self):
        d = dict(self.named_parameters())
        return [d[k] for k in self._order()]

    def flat_parameters(self):
        """One contiguous fp32 vector holding every parameter in the library's order; the nn.Parameters are
        re-pointed to views of it (re-done whenever `.to()` / `load_state_dict` broke the aliasing)."""
        ps = self._named()
        flat = self._flat
        ok = flat is not None and flat.device == ps[0].device
        if ok:
            base = flat.data_ptr()
            ok = all(p.data_ptr() == base + 4 * o and p.is_contiguous() for p, o in zip(ps, self._offsets))
        if not ok:
            flat = torch.empty(self._offsets[16], device=ps[0].device, dtype=torch.float32)
            for p, o in zip(ps, self._offsets):
                flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat[o:o + p.numel()].view(p.shape)
            self._flat = flat
        return flat

    def workspace(self, B):
        dev = self.pool_w.device
        key = (B, dev)
        if key not in self._ws:
            self._ws = {key: torch.empty(lib.workspace_bytes(self.shape, B) // 4, device=dev, dtype=torch.float32)}
        return self._ws[key]

    # ---- reference call signature ------------------------------------------------------------------
    def forward(self, ms, pan=None):
        """`net(ms, pan)` (mainsolver.py:52); with gmf.single_input also `net(data)` (tostagesolver.py:274)."""
        if not ms.is_cuda:
            raise lib.DmfError('model.gmfnet.Net runs on the GPU only (hand-written HIP kernels); '
                               'set cfg["device"] to "cuda:0"')
        if pan is None and not self.arch['single_input']:
            raise TypeError('forward(x) with one input needs cfg["gmf"]["single_input"] = 1')
        lib.shape_supported(self.shape)
        a = ms.contiguous().float()
        if pan is None:
            b = lib.band_mean_patches(a)
        else:
            b = pan.contiguous().float()
        theta = self.flat_parameters()
        if self.arch['attention']:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                theta_g = torch.cat([p.reshape(-1) for p in self._named()])
                return _GmfAttnFunction.apply(self, a, b, theta_g)
            inp = lib.input_patches(self.shape, a, b, half=self.arch.get('half', 0))
            B = a.shape[0]
            logits = torch.empty(B, self.arch['K'], device=a.device, dtype=torch.float32)
            ws = torch.empty(lib.attn_workspace_bytes(self.shape, B), device=a.device, dtype=torch.uint8)
            lib.forward_attn(self.shape, inp, theta, self.pool_w, ws, logits)
            return logits
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # route the flat gradient back to the individual parameters through a differentiable cat
            theta_g = torch.cat([p.reshape(-1) for p in self._named()])
            return _GmfFunction.apply(self, a, b, theta_g)
        inp = lib.input_patches(self.shape, a, b, half=self.arch.get('half', 0))
        logits = torch.empty(a.shape[0], self.arch['K'], device=a.device, dtype=torch.float32)
        lib.forward(self.shape, inp, theta, self.pool_w, logits)
        return logits
