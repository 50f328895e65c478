"""GPU parity: the HIP path (through the C ABI, dmf/lib.py) against the CPU oracle (oracle/gmfnet_ref.py).

Tolerances (fp32, BASELINE.json north_star): logits <= 1e-5 absolute; gradients <= 1e-5 absolute +
1e-4 relative (different but fixed summation orders over 121 pixels x B patches).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'dual-modal-fusion_amd')

SHAPES = {
    # name: (C, C2, P, S, K)
    'tiny': (8, 1, 5, 4, 5),
    'tiny1': (8, 1, 5, 1, 5),
    'hsi': (200, 1, 11, 1, 17),
    'hsi224': (224, 3, 11, 1, 17),
    'panms': (4, 1, 16, 4, 12),
    'qua': (4, 1, 16, 1, 12),        # stage 2 of the two-stage path: one 4-band stream + its band mean
    'quatiny': (4, 1, 5, 1, 5),
    # rows of the v2 shape table (csrc/dmf_patch_v2.hip, DMF_V2_SHAPES): other patch sizes of the two HSI nets
    'hsi9': (200, 1, 9, 1, 17),
    'hsi7': (200, 1, 7, 1, 17),
    'hsi224p9': (224, 3, 9, 1, 17),
    # NOT in the compiled table: added at build time from tests/extra_shapes.txt (build.py --shapes), see
    # test_row_added_at_build_time
    'extra7': (8, 1, 7, 1, 5),
}


def make_cfg(name):
    C, C2, P, S, K = SHAPES[name]
    return {'patch_size': P, 'Categories_Number': K, 'data_city': 's', 'DATA_DICT': {'s': {'size': [64, 64, C]}},
            'scale': S, 'aux_bands': C2,
            'gmf': {'width': 32 if name in ('hsi224', 'hsi224p9') else 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}


def nets(name, seed=0):
    from oracle.gmfnet_ref import Net as RefNet
    from model.gmfnet import Net as HipNet
    cfg = make_cfg(name)
    torch.manual_seed(seed)
    ref = RefNet(cfg)
    # make biases / weights less symmetric than the default init so every path carries signal
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    hip = HipNet(cfg)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to('cuda:0')
    return cfg, ref, hip


def rand_batch(name, B, seed=1):
    C, C2, P, S, K = SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, C, P, P, generator=g)
    b = torch.rand(B, C2, S * P, S * P, generator=g)
    t = torch.randint(0, K, (B,), generator=g)
    return a, b, t


def ref_grads(ref, a, b, t, scale=None):
    ref.zero_grad()
    logits = ref(a, b)
    loss = torch.nn.functional.cross_entropy(logits, t)
    loss.backward()
    return logits.detach(), loss.item(), {k: p.grad.detach().clone() for k, p in ref.named_parameters()}


def assert_close(got, want, atol, rtol, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), '%s: %d/%d out of tolerance, max abs err %.3e (|ref| max %.3e) at %s' % (
        what, int(bad.sum()), bad.numel(), err.max().item(), want.abs().max().item(),
        np.unravel_index(int(err.argmax()), tuple(err.shape)) if err.dim() else ())


ALL = ['tiny', 'tiny1', 'hsi', 'hsi224', 'panms', 'qua', 'quatiny', 'hsi9', 'hsi7', 'hsi224p9']


@pytest.mark.parametrize('name', ALL)
@pytest.mark.parametrize('B', [1, 37, 300])
def test_forward_patches(name, B):
    if name in ('hsi224', 'panms', 'hsi224p9') and B == 300:
        B = 260
    cfg, ref, hip = nets(name)
    a, b, t = rand_batch(name, B)
    with torch.no_grad():
        want = ref(a, b)
        got = hip(a.cuda(), b.cuda())
    assert got.shape == (B, SHAPES[name][4])
    assert_close(got, want, 1e-5, 0, 'logits[%s,B=%d]' % (name, B))


def _scene(name, H=23, W=19, seed=5):
    C, C2, P, S, K = SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    A = torch.rand(H + P - 1, W + P - 1, C, generator=g)
    Bm = torch.rand(S * (H + P - 1), S * (W + P - 1), C2, generator=g)
    return A, Bm


@pytest.mark.parametrize('name', ALL)
def test_forward_gather_and_pred(name):
    from dmf import lib
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    H, W = 23, 19
    A, Bm = _scene(name, H, W)
    g = torch.Generator().manual_seed(2)
    Bn = 301
    xy = torch.stack([torch.randint(0, H, (Bn,), generator=g), torch.randint(0, W, (Bn,), generator=g)], 1).int()
    xy[0] = torch.tensor([0, 0]); xy[1] = torch.tensor([H - 1, W - 1])          # extremes of the padded scene
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    b = torch.stack([Bm[S * x:S * x + S * P, S * y:S * y + S * P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    with torch.no_grad():
        want = ref(a, b)
    Ad, Bd, xyd = A.cuda(), Bm.cuda(), xy.cuda()
    lib.check_xy_bounds(hip.shape, Ad, Bd, xy.numpy())
    inp = lib.input_gather(hip.shape, Ad, Bd, xyd)
    logits = torch.empty(Bn, K, device='cuda')
    pred = torch.empty(Bn, dtype=torch.int32, device='cuda')
    lib.forward(hip.shape, inp, hip.flat_parameters(), hip.pool_w, logits, pred)
    assert_close(logits, want, 1e-5, 0, 'gather logits[%s]' % name)
    assert torch.equal(pred.cpu().long(), logits.cpu().argmax(1))
    # class map must match the oracle wherever the oracle's top-2 margin exceeds the tolerance
    top2 = want.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(pred.cpu().long()[safe], want.argmax(1)[safe])


@pytest.mark.parametrize('name', ALL)
@pytest.mark.parametrize('B', [3, 64, 300])
def test_train_fwd_bwd_grads(name, B):
    from dmf import lib
    if name in ('hsi224', 'panms', 'hsi224p9') and B == 300:
        B = 260
    cfg, ref, hip = nets(name)
    a, b, t = rand_batch(name, B)
    want_logits, want_loss, want_g = ref_grads(ref, a, b, t)
    K = SHAPES[name][4]
    ad, bd = a.cuda(), b.cuda()          # keep alive: the input descriptor holds raw device pointers
    inp = lib.input_patches(hip.shape, ad, bd)
    theta = hip.flat_parameters()
    logits = torch.empty(B, K, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = hip.workspace(B)
    lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), 1.0 / B, logits, loss, ws)
    grad = torch.empty_like(theta)
    lib.grad_reduce(hip.shape, B, ws, grad)
    torch.cuda.synchronize()
    assert_close(logits, want_logits, 1e-5, 0, 'train logits')
    assert abs(loss.mean().item() - want_loss) < 1e-5
    from model.gmfnet import PARAM_ORDER
    off = hip._offsets
    for i, k in enumerate(PARAM_ORDER):
        g = grad[off[i]:off[i] + want_g[k].numel()].view(want_g[k].shape)
        assert_close(g, want_g[k], 1e-5, 1e-4, 'grad %s [%s,B=%d]' % (k, name, B))


@pytest.mark.parametrize('name,B', [('hsi', 700), ('qua', 1100), ('panms', 600)])
def test_train_grads_three_and_more_patches_per_workgroup(name, B):
    """Batches beyond 512: a workgroup walks three or more patches (the later-patch path of the v2 kernel — aux rows by direct
    loads, no barrier X, slab row accumulated in place), in gather mode, against the oracle."""
    from dmf import lib
    from model.gmfnet import PARAM_ORDER
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    H, W = 23, 19
    A, Bm = _scene(name, H, W)
    g = torch.Generator().manual_seed(13)
    xy = torch.stack([torch.randint(0, H, (B,), generator=g), torch.randint(0, W, (B,), generator=g)], 1).int()
    t = torch.randint(0, K, (B,), generator=g)
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    b = torch.stack([Bm[S * x:S * x + S * P, S * y:S * y + S * P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    want_logits, want_loss, want_g = ref_grads(ref, a, b, t)
    Ad, Bd, xyd = A.cuda(), Bm.cuda(), xy.cuda()
    inp = lib.input_gather(hip.shape, Ad, Bd, xyd)
    theta = hip.flat_parameters()
    logits = torch.empty(B, K, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = hip.workspace(B)
    lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), 1.0 / B, logits, loss, ws)
    grad = torch.empty_like(theta)
    lib.grad_reduce(hip.shape, B, ws, grad)
    assert_close(logits, want_logits, 1e-5, 0, 'logits [%s, B=%d]' % (name, B))
    assert abs(loss.mean().item() - want_loss) < 1e-5
    off = hip._offsets
    for i, k in enumerate(PARAM_ORDER):
        gk = grad[off[i]:off[i] + want_g[k].numel()].view(want_g[k].shape)
        assert_close(gk, want_g[k], 1e-5, 1e-4, 'grad %s [%s, B=%d]' % (k, name, B))


@pytest.mark.parametrize('name', ['panms', 'tiny'])
@pytest.mark.parametrize('extra', [1, 3])
def test_aux_scene_pitch_not_a_multiple_of_four(name, extra):
    """The reference pads the 2-D aux image by 4p - 1 (function.py:99-117): a 1024-wide PAN becomes 1087 wide, so the rows of the
    resident aux scene are only 4-byte aligned.  The v2 kernel's 16-byte aux fetches must take that (forward, loss, gradients
    against the oracle, gather mode, patches at the scene's far corner included)."""
    from dmf import lib
    from model.gmfnet import PARAM_ORDER
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    H, W, B = 23, 19, 150
    g = torch.Generator().manual_seed(17)
    A = torch.rand(H + P - 1, W + P - 1, C, generator=g)
    Bm = torch.rand(S * (H + P - 1) + extra, S * (W + P - 1) + extra, C2, generator=g)      # pitch % 4 == extra
    xy = torch.stack([torch.randint(0, H, (B,), generator=g), torch.randint(0, W, (B,), generator=g)], 1).int()
    xy[0] = torch.tensor([H - 1, W - 1]); xy[1] = torch.tensor([0, 0])
    t = torch.randint(0, K, (B,), generator=g)
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    b = torch.stack([Bm[S * x:S * x + S * P, S * y:S * y + S * P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    want_logits, want_loss, want_g = ref_grads(ref, a, b, t)
    Ad, Bd, xyd = A.cuda(), Bm.cuda(), xy.cuda()
    assert (Bd.shape[1] * C2) % 4 == extra
    inp = lib.input_gather(hip.shape, Ad, Bd, xyd)
    theta = hip.flat_parameters()
    logits = torch.empty(B, K, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = hip.workspace(B)
    lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), 1.0 / B, logits, loss, ws)
    grad = torch.empty_like(theta)
    lib.grad_reduce(hip.shape, B, ws, grad)
    assert_close(logits, want_logits, 1e-5, 0, 'logits [%s, pitch %% 4 = %d]' % (name, extra))
    assert abs(loss.mean().item() - want_loss) < 1e-5
    off = hip._offsets
    for i, k in enumerate(PARAM_ORDER):
        gk = grad[off[i]:off[i] + want_g[k].numel()].view(want_g[k].shape)
        assert_close(gk, want_g[k], 1e-5, 1e-4, 'grad %s [%s, pitch %% 4 = %d]' % (k, name, extra))
    fl = torch.empty(B, K, device='cuda')
    lib.forward(hip.shape, inp, theta, hip.pool_w, fl)
    assert_close(fl, want_logits, 1e-5, 0, 'eval logits [%s, pitch %% 4 = %d]' % (name, extra))


@pytest.mark.parametrize('name', ['tiny', 'hsi', 'hsi224'])
def test_train_gather_equals_patches(name):
    """Both input modes feed the same arithmetic: bit-identical logits, loss and gradient."""
    from dmf import lib
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    H, W = 23, 19
    A, Bm = _scene(name, H, W)
    g = torch.Generator().manual_seed(3)
    Bn = 130
    xy = torch.stack([torch.randint(0, H, (Bn,), generator=g), torch.randint(0, W, (Bn,), generator=g)], 1).int()
    t = torch.randint(0, K, (Bn,), generator=g).int().cuda()
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()]).contiguous().cuda()
    b = torch.stack([Bm[S * x:S * x + S * P, S * y:S * y + S * P, :].permute(2, 0, 1) for x, y in xy.tolist()]).contiguous().cuda()
    theta = hip.flat_parameters()
    outs = []
    Ad, Bd, xyd = A.cuda(), Bm.cuda(), xy.cuda()
    for inp in (lib.input_patches(hip.shape, a, b), lib.input_gather(hip.shape, Ad, Bd, xyd)):
        logits = torch.empty(Bn, K, device='cuda'); loss = torch.empty(Bn, device='cuda')
        ws = torch.zeros(lib.workspace_bytes(hip.shape, Bn) // 4, device='cuda')
        lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t, 1.0 / Bn, logits, loss, ws)
        grad = torch.empty_like(theta)
        lib.grad_reduce(hip.shape, Bn, ws, grad)
        outs.append((logits.cpu(), loss.cpu(), grad.cpu()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize('name', ['tiny1', 'hsi', 'qua', 'quatiny', 'hsi9'])
@pytest.mark.parametrize('B', [5, 300])
def test_unit_gradient_step_equals_autograd(name, B):
    """The two-launch step for batch-coupled losses: dmf_forward_unit (forward + unit gradients per patch), a caller-made
    dL/dlogits, dmf_backward_unit — against torch autograd of the oracle for the same upstream gradient, in gather mode."""
    from dmf import lib
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    assert lib.unit_supported(hip.shape)
    H, W = 23, 19
    A, Bm = _scene(name, H, W)
    g = torch.Generator().manual_seed(11)
    xy = torch.stack([torch.randint(0, H, (B,), generator=g), torch.randint(0, W, (B,), generator=g)], 1).int()
    dl = torch.randn(B, K, generator=g) / B
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    b = torch.stack([Bm[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    ref.zero_grad()
    want_logits = ref(a, b)
    want_logits.backward(dl)
    want_g = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    Ad, Bd, xyd = A.cuda(), Bm.cuda(), xy.cuda()
    inp = lib.input_gather(hip.shape, Ad, Bd, xyd)
    theta = hip.flat_parameters()
    logits = torch.empty(B, K, device='cuda')
    ws = hip.workspace(B)
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    lib.forward_unit(hip.shape, inp, theta, hip.pool_w, logits, ws, adam_step_dev=step)
    lib.backward_unit(hip.shape, B, theta, dl.cuda(), ws)
    grad = torch.empty_like(theta)
    lib.grad_reduce(hip.shape, B, ws, grad)
    torch.cuda.synchronize()
    assert int(step.item()) == 1
    assert_close(logits, want_logits.detach(), 1e-5, 0, 'unit-step logits')
    from model.gmfnet import PARAM_ORDER
    off = hip._offsets
    for i, k in enumerate(PARAM_ORDER):
        gk = grad[off[i]:off[i] + want_g[k].numel()].view(want_g[k].shape)
        assert_close(gk, want_g[k], 1e-5, 1e-4, 'unit-step grad %s [%s,B=%d]' % (k, name, B))


@pytest.mark.parametrize('name', ['tiny', 'hsi'])
def test_autograd_boundary(name):
    """The reference solver's own step (mainsolver.py:51-55) through model.gmfnet.Net: torch CE + backward + Adam."""
    cfg, ref, hip = nets(name)
    B = 48
    a, b, t = rand_batch(name, B)
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt_h = torch.optim.Adam(hip.parameters(), lr=1e-3)
    ce = torch.nn.CrossEntropyLoss()
    for step in range(3):
        opt_r.zero_grad(); lr_ = ce(ref(a, b), t); lr_.backward(); opt_r.step()
        opt_h.zero_grad(); lh = ce(hip(a.cuda(), b.cuda()), t.cuda()); lh.backward(); opt_h.step()
        assert abs(lr_.item() - lh.item()) < 1e-5
    sd_r, sd_h = ref.state_dict(), hip.state_dict()
    assert list(sd_r.keys()) == list(sd_h.keys())
    for k in sd_r:
        assert_close(sd_h[k], sd_r[k], 2e-5, 1e-4, 'param %s after 3 Adam steps' % k)


def test_fused_adam_matches_torch_adam():
    """dmf_grad_reduce_adam == dmf_grad_reduce + torch.optim.Adam (utils/utils.py:10-12 defaults)."""
    from dmf import lib
    from oracle.gmfnet_ref import adam_step_ref
    cfg, ref, hip = nets('tiny')
    B = 64
    a, b, t = rand_batch('tiny', B)
    ad, bd = a.cuda(), b.cuda()
    inp = lib.input_patches(hip.shape, ad, bd)
    theta = hip.flat_parameters().clone()
    K = SHAPES['tiny'][4]
    m = torch.zeros_like(theta); v = torch.zeros_like(theta)
    th_ref = theta.cpu().clone(); m_ref = torch.zeros_like(th_ref); v_ref = torch.zeros_like(th_ref)
    logits = torch.empty(B, K, device='cuda'); loss = torch.empty(B, device='cuda')
    ws = torch.empty(lib.workspace_bytes(hip.shape, B) // 4, device='cuda')
    for step in range(1, 6):
        lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), 1.0 / B, logits, loss, ws)
        grad = torch.empty_like(theta)
        lib.grad_reduce_adam(hip.shape, B, ws, theta, m, v, grad, 1e-3, 0.9, 0.999, 1e-8, step)
        adam_step_ref(th_ref, grad.cpu(), m_ref, v_ref, step)
        assert_close(theta, th_ref, 1e-7, 1e-6, 'theta after fused Adam step %d' % step)
        th_ref = theta.cpu().clone(); m_ref = m.cpu().clone(); v_ref = v.cpu().clone()
    # separate Adam entry point, with gradient scaling (the post-all-reduce 1/world_size)
    th2 = theta.clone(); m2 = m.clone(); v2 = v.clone()
    g = torch.randn_like(theta)
    lib.adam_step(th2, g, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 6, 0.5)
    th_c, m_c, v_c = theta.cpu().clone(), m.cpu().clone(), v.cpu().clone()
    adam_step_ref(th_c, g.cpu() * 0.5, m_c, v_c, 6)
    assert_close(th2, th_c, 1e-7, 1e-6, 'dmf_adam_step')


def test_confusion_labelmap_pan2ms(golden_dir):
    from dmf import lib
    from oracle import datapath_ref as d
    g = torch.Generator().manual_seed(0)
    K, B = 17, 5000
    pred = torch.randint(0, K, (B,), generator=g).int(); tgt = torch.randint(0, K, (B,), generator=g).int()
    mat = torch.zeros(K, K, dtype=torch.int64, device='cuda')
    lib.confusion_accum(pred.cuda(), tgt.cuda(), K, mat)
    lib.confusion_accum(pred.cuda()[:100], tgt.cuda()[:100], K, mat)
    want = d.confusion(pred.numpy(), tgt.numpy(), K) + d.confusion(pred.numpy()[:100], tgt.numpy()[:100], K)
    assert np.array_equal(mat.cpu().numpy().astype(np.float64), want)
    H, W = 31, 17
    xy = torch.stack([torch.arange(H * W) // W, torch.arange(H * W) % W], 1).int()
    p2 = torch.randint(0, K, (H * W,), generator=g).int()
    lm = torch.full((H, W), -1, dtype=torch.int32, device='cuda')
    lib.labelmap_write(p2.cuda(), xy.cuda(), W, lm)
    assert torch.equal(lm.cpu().reshape(-1), p2)
    gg = np.load(os.path.join(golden_dir, 'g7_ihs.npz'))
    for pan_key, out_key in (('pan', 'p2m'), ('pan_r', 'p2m_r')):
        pan = torch.from_numpy(gg[pan_key]).cuda()
        Ho, Wo = gg[out_key].shape[:2]
        out = torch.empty(Ho, Wo, 4, dtype=torch.float64, device='cuda')
        lib.pan2ms(pan, Ho, Wo, out)
        assert np.array_equal(out.cpu().numpy(), gg[out_key]), 'pan2ms must be bit-exact (fp64, integer-like op)'


def test_unsupported_shape_fails_loudly():
    from dmf import lib
    from model.gmfnet import Net
    cfg = make_cfg('tiny'); cfg['patch_size'] = 7
    net = Net(cfg).cuda()
    with pytest.raises(lib.DmfError, match='no compiled kernel instance'):
        net(torch.zeros(1, 8, 7, 7).cuda(), torch.zeros(1, 1, 28, 28).cuda())
    with pytest.raises(lib.DmfError, match='GPU only'):
        Net(make_cfg('tiny'))(torch.zeros(1, 8, 5, 5), torch.zeros(1, 1, 20, 20))


def test_edge_cases_at_the_c_abi():
    """Empty batches are no-ops, malformed requests fail with a message (never a fault on the GPU): wrong tensor
    shapes, coordinates outside the padded scene, labels out of range, K above DMF_KMAX, attention nets sent to the
    conv-only entry points, plan lengths that are not whole batches."""
    from dmf import lib
    from dmf.engine import Scene, TrainEngine, EvalEngine
    cfg, ref, hip = nets('tiny1')
    C, C2, P, S, K = SHAPES['tiny1']
    theta = hip.flat_parameters()
    # B = 0: forward and the train step return without launching
    a0, b0 = torch.zeros(0, C, P, P, device='cuda'), torch.zeros(0, C2, P, P, device='cuda')
    lib.forward(hip.shape, lib.input_patches(hip.shape, a0, b0), theta, hip.pool_w, torch.empty(0, K, device='cuda'))
    assert hip(a0, b0).shape == (0, K)
    # wrong patch shape
    with pytest.raises(lib.DmfError, match='do not match'):
        lib.input_patches(hip.shape, torch.zeros(2, C, P + 1, P, device='cuda'), torch.zeros(2, C2, P, P, device='cuda'))
    # coordinates outside the scene are refused on the host
    A, Bm = _scene('tiny1', 9, 9)
    scene = Scene(A.numpy(), Bm.numpy(), 'cuda:0')
    ev = EvalEngine(hip, scene, 8)
    with pytest.raises(lib.DmfError, match='outside the padded scene'):
        ev.confusion(np.array([[9, 0]], dtype=np.int32), np.array([1], dtype=np.int32))
    with pytest.raises(lib.DmfError, match='outside the padded scene'):
        ev.label_map(np.array([[0, -1]], dtype=np.int32), 9, 9)
    eng = TrainEngine(hip, scene, 4, lr=1e-3)
    with pytest.raises(lib.DmfError, match='multiple of the batch size'):
        eng.load_plan(np.zeros((6, 2), dtype=np.int32), np.ones(6, dtype=np.int32))
    with pytest.raises(lib.DmfError, match='label outside'):
        eng.load_plan(np.zeros((4, 2), dtype=np.int32), np.full(4, K, dtype=np.int32))
    with pytest.raises(lib.DmfError, match='at most'):
        eng.step(torch.zeros(5, 2, dtype=torch.int32, device='cuda'), torch.ones(5, dtype=torch.int32, device='cuda'))
    # K above DMF_KMAX
    big = lib.make_shape(dict(hip.arch, K=65))
    with pytest.raises(lib.DmfError):
        lib.shape_supported(big)
    # attention nets must use their own entry points
    cfg_a, ref_a, hip_a = _attn_nets('tiny1')
    a, b, t = rand_batch('tiny1', 3)
    inp = lib.input_patches(hip_a.shape, a.cuda(), b.cuda())
    with pytest.raises(lib.DmfError, match='attention'):
        lib.forward(hip_a.shape, inp, hip_a.flat_parameters(), hip_a.pool_w, torch.empty(3, K, device='cuda'))
    ws = torch.empty(lib.workspace_bytes(hip_a.shape, 3) // 4, device='cuda')
    with pytest.raises(lib.DmfError, match='attention'):
        lib.train_fwd_bwd(hip_a.shape, inp, hip_a.flat_parameters(), hip_a.pool_w, t.int().cuda(), 1.0, torch.empty(3, K, device='cuda'),
                          torch.empty(3, device='cuda'), ws)
    aws = torch.empty(lib.attn_train_workspace_bytes(hip_a.shape, 3), dtype=torch.uint8, device='cuda')
    with pytest.raises(lib.DmfError, match='exactly one'):
        lib.train_attn_fwd_bwd(hip_a.shape, inp, hip_a.flat_parameters(), hip_a.pool_w, None, None, 1.0,
                               torch.empty(3, K, device='cuda'), None, ws, aws)
    # qua_loss: logits must be [4*bs, K]
    with pytest.raises(lib.DmfError, match='4\\*bs'):
        lib.qua_loss(torch.zeros(7, K, device='cuda'), 2, torch.zeros(2, dtype=torch.int32, device='cuda'),
                     lib.QuaParams(alpha=0.1, beta=0.05, gamma=1.0, epsilon=1e-8, tao=0.1))


# ------------------------------------------------------------------------------------------------ attention (forward)
def _attn_nets(name, seed=0):
    from oracle.gmfnet_ref import Net as RefNet
    from model.gmfnet import Net as HipNet
    cfg = make_cfg(name)
    cfg['gmf'] = dict(cfg['gmf'], attention=1)
    cfg['trans'] = {'embed_dim': 96, 'num_head': 3}
    torch.manual_seed(seed)
    ref = RefNet(cfg)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            p.add_(0.05 * torch.randn_like(p))
            if k.startswith('attn_'):
                p.mul_(3.0)                       # make the attention path carry real signal
    hip = HipNet(cfg)
    hip.load_state_dict(ref.state_dict())
    return cfg, ref, hip.to('cuda:0')


@pytest.mark.parametrize('name,B', [('tiny1', 1), ('tiny1', 70), ('tiny1', 300), ('hsi', 1), ('hsi', 70), ('hsi', 300),
                                    ('hsi', 1024)])      # 1024: the batch BASELINE configs[2] is quoted on (4 patches per CU)
def test_attention_forward(name, B):
    """Cross-modal attention forward (bf16 MFMA operands, fp32 accumulate) against the oracle that rounds the same
    operands to bf16.  Tolerance 2e-4 on logits (measured <= 3e-5): a bf16 operand that sits on a rounding boundary
    can round the other way when the fp32 accumulation order differs (SURVEY §7 'bf16 MFMA attention vs 1e-5'), while
    an fp32 attention differs from the bf16 one by ~2e-3; the fp32-only network is held to 1e-5 by the tests above."""
    cfg, ref, hip = _attn_nets(name)
    a, b, t = rand_batch(name, B)
    with torch.no_grad():
        want = ref(a, b)
        got = hip(a.cuda(), b.cuda())
        ref.arch['mfma_bf16'] = 0
        want_fp32 = ref(a, b)
    # the attention path must matter in this test, otherwise it proves nothing
    no_attn = (want_fp32 - want).abs().max().item()
    assert_close(got, want, 2e-4, 0, 'attention logits[%s,B=%d]' % (name, B))
    assert no_attn > 5e-4
    err = (got.cpu() - want).abs().max().item()
    print('attention %s B=%d: max|hip - bf16 oracle| = %.2e, |fp32 oracle - bf16 oracle| = %.2e' % (name, B, err, no_attn))


@pytest.mark.parametrize('name,B', [('tiny1', 2), ('tiny1', 70), ('tiny1', 300), ('hsi', 2), ('hsi', 70), ('hsi', 300),
                                    ('hsi', 1024)])      # 1024: BASELINE configs[2]'s batch
def test_attention_train_grads(name, B):
    """Fused fwd + CE + bwd of the attention network (token kernel, attention fwd+bwd kernel, dense conv backward,
    gradient reduce) against torch autograd through the oracle (bf16 forward operands, straight-through roundings).
    Tolerance: logits as in test_attention_forward; gradients 2e-5 absolute + 2e-3 relative.  Looser than the fp32
    network's 1e-4 because a bf16 operand on a rounding boundary may round the other way between the two forward
    passes (accumulation order), which moves the forward value by one bf16 ulp of one operand."""
    from dmf import lib
    cfg, ref, hip = _attn_nets(name)
    a, b, t = rand_batch(name, B)
    ref.zero_grad()
    want_logits = ref(a, b)
    loss = torch.nn.functional.cross_entropy(want_logits, t)
    loss.backward()
    K = SHAPES[name][4]
    ad, bd = a.cuda(), b.cuda()
    inp = lib.input_patches(hip.shape, ad, bd)
    theta = hip.flat_parameters()
    logits = torch.empty(B, K, device='cuda')
    lossv = torch.empty(B, device='cuda')
    ws = torch.empty(lib.workspace_bytes(hip.shape, B) // 4, device='cuda')
    aws = torch.empty(lib.attn_train_workspace_bytes(hip.shape, B), dtype=torch.uint8, device='cuda')
    lib.train_attn_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), None, 1.0 / B, logits, lossv, ws, aws)
    grad = torch.empty_like(theta)
    lib.grad_reduce(hip.shape, B, ws, grad)
    assert_close(logits, want_logits, 2e-4, 0, 'attention train logits[%s,B=%d]' % (name, B))
    assert abs(lossv.mean().item() - loss.item()) < 2e-4
    off = hip._offsets
    worst = 0.0
    for i, (k, p) in enumerate(zip(hip._order(), hip._named())):
        want = dict(ref.named_parameters())[k].grad
        got = grad[off[i]:off[i] + p.numel()].view(p.shape).cpu()
        err = (got - want).abs().max().item()
        worst = max(worst, err / (want.abs().max().item() + 1e-12))
        assert_close(got, want, 2e-5, 2e-3, 'grad %s[%s,B=%d]' % (k, name, B))
        if k.startswith('attn_'):
            assert want.abs().max().item() > 1e-6, 'attention gradient vanishes: the test would prove nothing'
    print('attention train %s B=%d: worst relative-to-max gradient error %.2e' % (name, B, worst))


def test_attention_autograd_and_engine_steps():
    """`net(ms, pan)` + torch loss + backward with attention on (drop-in path), and three fused engine steps against
    torch Adam on the oracle."""
    from dmf.engine import Scene, TrainEngine
    from oracle import solver_ref
    name = 'tiny1'
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = _attn_nets(name)
    a, b, t = rand_batch(name, 40)
    ref.zero_grad()
    torch.nn.functional.cross_entropy(ref(a, b), t).backward()
    hip.zero_grad()
    torch.nn.functional.cross_entropy(hip(a.cuda(), b.cuda()), t.cuda()).backward()
    for (k, pr), (_, ph) in zip(ref.named_parameters(), hip.named_parameters()):
        assert_close(ph.grad, pr.grad, 2e-5, 2e-3, 'autograd grad ' + k)
    # engine: resident scene, 3 steps of 32 patches, lr 1e-2
    H, W, Bn = 23, 19, 32
    A, Bm = _scene(name, H, W)
    g = torch.Generator().manual_seed(8)
    xy = torch.stack([torch.randint(0, H, (3 * Bn,), generator=g), torch.randint(0, W, (3 * Bn,), generator=g)], 1).int()
    lab = torch.randint(0, K, (3 * Bn,), generator=g).int()
    eng = TrainEngine(hip, Scene(A.numpy(), Bm.numpy(), 'cuda:0'), Bn, lr=1e-2)
    eng.load_plan(xy, lab)
    eng.run_plan(3, 0)
    got_losses = eng.mean_losses().numpy()
    want_losses, _ = solver_ref.train_steps(ref, A.numpy(), Bm.numpy(), xy.numpy(), lab.numpy(), Bn, P, S, lr=1e-2)
    print('attention engine losses', got_losses, want_losses)
    assert np.abs(got_losses - np.array(want_losses)).max() < 5e-4
    # ADAM's first steps move every element by ~lr whatever the size of its gradient, so an element whose gradient is
    # rounding noise can go either way; the bulk must agree closely, the tail is bounded by steps * lr
    d = torch.cat([(ph.detach().cpu() - pr.detach()).abs().reshape(-1)
                   for (_, pr), (_, ph) in zip(ref.named_parameters(), hip.named_parameters())]).numpy()
    print('attention engine parameters after 3 steps: 99th percentile diff %.2e, max %.2e' % (np.percentile(d, 99), d.max()))
    assert np.percentile(d, 99) < 2e-4 and d.max() <= 2 * 3 * 1e-2 + 1e-6


@pytest.mark.parametrize('name', ['hsi', 'tiny1', 'panms'])
def test_forward_ce_matches_torch_cross_entropy(name):
    """dmf_forward_ce (the validation pass's launch: evaluation forward + per-patch cross-entropy) against
    torch.nn.functional.cross_entropy on the oracle's logits, labels at both ends of the class range included."""
    from dmf import lib
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    H, W = 23, 19
    A, Bm = _scene(name, H, W)
    g = torch.Generator().manual_seed(4)
    Bn = 97
    xy = torch.stack([torch.randint(0, H, (Bn,), generator=g), torch.randint(0, W, (Bn,), generator=g)], 1).int()
    t = torch.randint(0, K, (Bn,), generator=g)
    t[0], t[1] = 0, K - 1
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    b = torch.stack([Bm[S * x:S * x + S * P, S * y:S * y + S * P, :].permute(2, 0, 1) for x, y in xy.tolist()])
    with torch.no_grad():
        want_logits = ref(a, b)
        want = torch.nn.functional.cross_entropy(want_logits, t, reduction='none')
    Ad, Bd, xyd = A.cuda(), Bm.cuda(), xy.cuda()
    inp = lib.input_gather(hip.shape, Ad, Bd, xyd)
    logits = torch.empty(Bn, K, device='cuda'); loss = torch.full((Bn,), float('nan'), device='cuda')
    pred = torch.empty(Bn, dtype=torch.int32, device='cuda')
    lib.forward_ce(hip.shape, inp, hip.flat_parameters(), hip.pool_w, t.int().cuda(), logits, loss, pred)
    assert_close(logits, want_logits, 1e-5, 0, 'forward_ce logits[%s]' % name)
    assert_close(loss, want, 2e-5, 0, 'forward_ce loss[%s]' % name)
    assert abs(loss.double().sum().item() - want.double().sum().item()) < 1e-4


# ---------------------------------------------------------------------------------------------- rows added at build time
EXTRA_LIB = os.path.join(PKG, 'dmf', 'libdmf_hip_extra.so')


if os.environ.get('DMF_TEST_EXTRA_ROW') == '1':        # (collected only in the child process of test_row_added_at_build_time)
    @pytest.mark.parametrize('B', [1, 37, 300])
    def test_extra_row_inner(B):
        """On the library that holds the extra row: forward, gather + argmax, fused train step against the oracle."""
        test_forward_patches('extra7', B)
        test_train_fwd_bwd_grads('extra7', B)
        if B == 1:
            test_forward_gather_and_pred('extra7')


def test_row_added_at_build_time():
    """`build.py --shapes FILE` (missing 3: a shape outside the compiled table used to be a dead end): the stock
    library refuses the 8-band / 7x7 shape by name, a library built with tests/extra_shapes.txt runs it, and forward, gather +
    argmax and the fused train step's gradients agree with the oracle like every stock row."""
    import subprocess
    import sys
    from dmf import lib
    from model.gmfnet import Net as HipNet
    with pytest.raises(lib.DmfError, match='no compiled kernel instance'):
        HipNet(make_cfg('extra7')).to('cuda:0')(*[x.cuda() for x in rand_batch('extra7', 2)[:2]])
    sys.path.insert(0, PKG)
    import build as dmf_build
    out = dmf_build.build(shapes=os.path.join(REPO, 'tests', 'extra_shapes.txt'), suffix='extra', verbose=False)
    assert out == EXTRA_LIB and os.path.exists(out)
    env = dict(os.environ, DMF_LIB=EXTRA_LIB, DMF_TEST_EXTRA_ROW='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'test_extra_row_inner', '-m', 'gpu',
                        '-p', 'no:cacheprovider'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and '3 passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
