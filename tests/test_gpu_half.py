"""GPU parity of the fp16 variant (cfg['gmf']['half'] = 1; include/dmf.h: dmf_input.half) and of the device-resident loss
scaler against the oracle RUN WITH THE SAME ROUNDINGS (oracle/gmfnet_ref.py::Net.branches): primary modality and
spec_a.weight rounded to fp16 (nearest even), exact products, fp32 accumulation; everything downstream fp32.

The reference uses mixed precision only in stage 1 (`autocast` + two `GradScaler`s, tostagesolver.py:83-84,98,119), whose
networks it does not ship — so no reference output exists for this path: parity unpinned beyond the oracle's definition.
The scaler's update rule is checked against torch.amp.GradScaler itself (CPU) driving the oracle net.

Tolerances: as the fp32 path (logits 1e-5 abs, gradients 1e-5 abs + 1e-4 rel) — the operands are identical fp16 values on
both sides, only the fp32 summation order differs.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import SHAPES, assert_close, make_cfg

pytestmark = pytest.mark.gpu

HALF = ['tiny1', 'hsi', 'hsi224', 'qua']       # rows of DMF_V2_HALF_SHAPES (csrc/dmf_patch_v2.hip)


def half_nets(name, seed=0):
    from oracle.gmfnet_ref import Net as RefNet
    from model.gmfnet import Net as HipNet
    cfg = make_cfg(name)
    cfg['gmf']['half'] = 1
    torch.manual_seed(seed)
    ref = RefNet(cfg)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    hip = HipNet(cfg)
    hip.load_state_dict(ref.state_dict())
    return cfg, ref, hip.to('cuda:0')


def scene(name, H=23, W=19, seed=5):
    """Random scenes; a sprinkle of fp16-SUBNORMAL magnitudes (|x| < 6.1e-5) and exact zeros in the primary one: the fp16
    instructions must not flush what numpy / torch keep."""
    C, C2, P, S, K = SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    A = torch.rand(H + P - 1, W + P - 1, C, generator=g) - 0.3
    tiny = torch.rand(A.shape, generator=g) < 0.02
    A[tiny] *= 4e-5
    A[torch.rand(A.shape, generator=g) < 0.01] = 0.0
    Bm = torch.rand(S * (H + P - 1), S * (W + P - 1), C2, generator=g)
    return A, Bm


def cut(A, Bm, xy, P, S):
    a = torch.stack([A[x:x + P, y:y + P, :].permute(2, 0, 1) for x, y in xy.tolist()]).contiguous()
    b = torch.stack([Bm[S * x:S * x + S * P, S * y:S * y + S * P, :].permute(2, 0, 1) for x, y in xy.tolist()]).contiguous()
    return a, b


def test_half_instances_and_refusals():
    from dmf import lib
    from model.gmfnet import Net
    for name in HALF:
        cfg, ref, hip = half_nets(name)
        assert lib.half_supported(hip.shape), name
    cfg = make_cfg('hsi9'); cfg['gmf']['half'] = 1
    net = Net(cfg).cuda()
    assert not lib.half_supported(net.shape)
    with pytest.raises(lib.DmfError, match='no fp16-scene kernel'):
        net(torch.zeros(1, 200, 9, 9).cuda(), torch.zeros(1, 1, 9, 9).cuda())
    # an fp16 scene for a shape without such a kernel must not reach the fp32 gather
    A = torch.zeros(20, 20, 200, dtype=torch.float16, device='cuda'); Bm = torch.zeros(20, 20, 1, device='cuda')
    xy = torch.zeros(1, 2, dtype=torch.int32, device='cuda')
    with pytest.raises(lib.DmfError, match='no fp16-scene kernel'):
        lib.forward(net.shape, lib.input_gather(net.shape, A, Bm, xy), net.flat_parameters(), net.pool_w,
                    torch.empty(1, 17, device='cuda'))


@pytest.mark.parametrize('name', HALF)
@pytest.mark.parametrize('B', [3, 300])
def test_half_train_step_against_oracle(name, B):
    """Forward, loss and every gradient, in both input modes (fp32 patches rounded while staged / fp16 resident scene):
    the two modes must agree bit for bit, and both with the oracle run with the same roundings."""
    from dmf import lib
    from model.gmfnet import PARAM_ORDER
    C, C2, P, S, K = SHAPES[name]
    if name == 'hsi224' and B == 300:
        B = 260
    cfg, ref, hip = half_nets(name)
    H, W = 23, 19
    A, Bm = scene(name, H, W)
    g = torch.Generator().manual_seed(3)
    xy = torch.stack([torch.randint(0, H, (B,), generator=g), torch.randint(0, W, (B,), generator=g)], 1).int()
    xy[0] = torch.tensor([H - 1, W - 1])
    t = torch.randint(0, K, (B,), generator=g)
    a, b = cut(A, Bm, xy, P, S)
    ref.zero_grad()
    want_logits = ref(a, b)
    loss = torch.nn.functional.cross_entropy(want_logits, t)
    loss.backward()
    want_g = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}

    theta = hip.flat_parameters()
    ad, bd = a.cuda(), b.cuda()
    Ah, Bd, xyd = A.cuda().to(torch.float16), Bm.cuda(), xy.cuda()
    outs = []
    for inp in (lib.input_patches(hip.shape, ad, bd, half=True), lib.input_gather(hip.shape, Ah, Bd, xyd)):
        assert inp.half == 1
        logits = torch.empty(B, K, device='cuda'); lossv = torch.empty(B, device='cuda')
        ws = torch.zeros(lib.workspace_bytes(hip.shape, B) // 4, device='cuda')
        lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), 1.0 / B, logits, lossv, ws)
        grad = torch.empty_like(theta)
        lib.grad_reduce(hip.shape, B, ws, grad)
        torch.cuda.synchronize()
        outs.append((logits.cpu(), lossv.cpu(), grad.cpu()))
        assert_close(logits, want_logits.detach(), 1e-5, 0, 'half logits[%s]' % name)
        assert abs(lossv.mean().item() - loss.item()) < 1e-5
        off = hip._offsets
        for i, k in enumerate(PARAM_ORDER):
            gk = grad[off[i]:off[i] + want_g[k].numel()].view(want_g[k].shape)
            assert_close(gk, want_g[k], 1e-5, 1e-4, 'half grad %s [%s,B=%d]' % (k, name, B))
        # eval twin
        fl = torch.empty(B, K, device='cuda'); pred = torch.empty(B, dtype=torch.int32, device='cuda')
        lib.forward(hip.shape, inp, theta, hip.pool_w, fl, pred)
        assert_close(fl, want_logits.detach(), 1e-5, 0, 'half eval logits[%s]' % name)
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize('name', ['tiny1', 'hsi'])
def test_half_rounding_is_what_separates_it_from_fp32(name):
    """With inputs large enough for the fp16 rounding to move the logits well beyond the fp32 tolerance, the HIP result
    sits on the half oracle, far from the fp32 oracle (i.e. the test above cannot pass on the fp32 kernel)."""
    from oracle.gmfnet_ref import Net as RefNet
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = half_nets(name)
    ref32 = RefNet(make_cfg(name)); ref32.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(40, C, P, P, generator=g) - 0.4) * 256.0
    b = torch.rand(40, C2, S * P, S * P, generator=g)
    with torch.no_grad():
        w16, w32 = ref(a, b), ref32(a, b)
        got = hip(a.cuda(), b.cuda()).cpu()
    gap = (w16 - w32).abs().max().item()
    err = (got - w16).abs().max().item()
    assert gap > 3e-4, gap
    assert err < 0.05 * gap, (err, gap)


@pytest.mark.parametrize('name', ['tiny1', 'qua'])
def test_half_unit_step_and_drop_in_boundary(name):
    """(a) the two-launch unit-gradient step on an fp16 scene against autograd of the oracle; (b) the plug-in boundary
    `Net(args=cfg)(ms, pan)` with gmf.half: forward + backward through torch autograd."""
    from dmf import lib
    from model.gmfnet import PARAM_ORDER
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = half_nets(name)
    B, H, W = 70, 23, 19
    A, Bm = scene(name, H, W)
    g = torch.Generator().manual_seed(11)
    xy = torch.stack([torch.randint(0, H, (B,), generator=g), torch.randint(0, W, (B,), generator=g)], 1).int()
    dl = torch.randn(B, K, generator=g) / B
    a, b = cut(A, Bm, xy, P, S)
    ref.zero_grad()
    want_logits = ref(a, b)
    want_logits.backward(dl)
    want_g = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    theta = hip.flat_parameters()
    Ah, Bd, xyd = A.cuda().to(torch.float16), Bm.cuda(), xy.cuda()
    inp = lib.input_gather(hip.shape, Ah, Bd, xyd)
    logits = torch.empty(B, K, device='cuda')
    ws = hip.workspace(B)
    lib.forward_unit(hip.shape, inp, theta, hip.pool_w, logits, ws)
    lib.backward_unit(hip.shape, B, theta, dl.cuda(), ws)
    grad = torch.empty_like(theta)
    lib.grad_reduce(hip.shape, B, ws, grad)
    assert_close(logits, want_logits.detach(), 1e-5, 0, 'half unit-step logits')
    off = hip._offsets
    for i, k in enumerate(PARAM_ORDER):
        gk = grad[off[i]:off[i] + want_g[k].numel()].view(want_g[k].shape)
        assert_close(gk, want_g[k], 1e-5, 1e-4, 'half unit-step grad %s' % k)
    # (b)
    hip.zero_grad()
    out = hip(a.cuda(), b.cuda())
    out.backward(dl.cuda())
    assert_close(out, want_logits.detach(), 1e-5, 0, 'drop-in logits (half)')
    for k, p in hip.named_parameters():
        assert_close(p.grad, want_g[k], 1e-5, 1e-4, 'drop-in grad %s (half)' % k)


def _plan(name, n_steps, B, H, W, seed, poison_steps=()):
    C, C2, P, S, K = SHAPES[name]
    A, Bm = scene(name, H, W, seed)
    A[H + P - 2, W + P - 2, 0] = float('inf')          # only a patch at (H-1, W-1) sees it
    g = torch.Generator().manual_seed(seed + 1)
    xy = torch.stack([torch.randint(0, H - 1, (n_steps * B,), generator=g), torch.randint(0, W - 1, (n_steps * B,), generator=g)], 1).int()
    for s in poison_steps:
        xy[s * B + 1] = torch.tensor([H - 1, W - 1])
    t = torch.randint(0, K, (n_steps * B,), generator=g)
    return A, Bm, xy, t


@pytest.mark.parametrize('graph', [0, 4])
@pytest.mark.parametrize('half', [0, 1])
def test_loss_scaler_follows_gradscaler(graph, half):
    """scale -> backward -> unscale -> skip-or-step -> update on the device (dmf_train_fwd_bwd_scaled + dmf_unscale_adam)
    against torch.amp.GradScaler driving the oracle net with torch.optim.Adam: same skipped steps (a poisoned patch makes
    the gradient non-finite), same scale trajectory, same parameters.  graph = 4: the steps replay from a captured graph."""
    from dmf import lib
    from dmf.engine import Scene, TrainEngine, LossScaler
    name = 'tiny1'
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = (half_nets(name) if half else __import__('test_gpu_parity').nets(name))
    n, B, H, W = 12, 16, 23, 19
    poison = (2, 3, 8)
    A, Bm, xy, t = _plan(name, n, B, H, W, 21, poison)
    if half:
        A[torch.isinf(A)] = 70000.0                    # beyond fp16's range: the fp16 scene itself carries the inf
    gs = torch.amp.GradScaler('cpu', init_scale=2.0 ** 12, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    scales, losses = [], []
    for s in range(n):
        a, b = cut(A, Bm, xy[s * B:(s + 1) * B], P, S)
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(ref(a, b), t[s * B:(s + 1) * B])
        gs.scale(loss).backward()
        gs.step(opt); gs.update()
        scales.append(gs.get_scale()); losses.append(loss.item())
    want_skipped = len(poison)

    sc = LossScaler('cuda:0', init_scale=2.0 ** 12, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    eng = TrainEngine(hip, Scene(A.numpy(), Bm.numpy(), 'cuda:0', half=bool(half)), B, lr=1e-3, scaler=sc)
    eng.load_plan(xy, t)
    got_scales = []
    if graph:
        for _ in range(n // graph):
            eng.run_plan(graph, steps_per_graph=graph)
            got_scales.append(sc.get_scale())
        assert got_scales == scales[graph - 1::graph]
    else:
        for _ in range(n):
            eng.run_plan(1)
            got_scales.append(sc.get_scale())
        assert got_scales == scales
    assert sc.skipped_steps() == want_skipped
    assert int(eng.dev_step.item()) == n - want_skipped          # skipped steps do not count for the bias corrections
    got_losses = eng.mean_losses().numpy()
    ok = np.isfinite(losses)
    # (poisoned steps: torch reports a NaN loss; the kernel's ReLU is v_med3 / fmaxf, which drop NaN, so its loss may be finite
    # while the gradient dW = dY * inf is not — the step is skipped on both sides, which is what is compared)
    assert np.allclose(got_losses[ok], np.asarray(losses)[ok], atol=2e-5)
    sd = ref.state_dict()
    for k, v in hip.state_dict().items():
        assert_close(v, sd[k], 3e-5, 1e-4, 'param %s after %d scaled steps' % (k, n))


def test_stage2_with_scaler_and_half_scene():
    """Stage 2 (qua_loss) on an fp16 tall scene with the loss scaler in the loop: finite steps reproduce the oracle's
    Adam trajectory (GradScaler + autograd of oracle qua_loss), from a captured graph."""
    from dmf.engine import QuaScene, QuaTrainEngine, LossScaler
    from oracle.solver_ref import materialise4
    from oracle import datapath_ref as dref
    name = 'qua'
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = half_nets(name)
    cfg['gmf']['single_input'] = 1
    from oracle.gmfnet_ref import Net as RefNet
    from model.gmfnet import Net as HipNet
    ref1 = RefNet(cfg); ref1.load_state_dict(ref.state_dict())
    hip1 = HipNet(cfg); hip1.load_state_dict(ref.state_dict()); hip1 = hip1.cuda()
    dqtl = {'alpha': 1.0, 'beta': 0.5, 'gamma': 0.5, 'epsilon': 1e-8, 'tao': 2.0}
    g = torch.Generator().manual_seed(4)
    H, W, bs, n = 20, 18, 8, 6
    scenes = [(torch.rand(H + P - 1, W + P - 1, C, generator=g) - 0.2).numpy() for _ in range(4)]
    xy = torch.stack([torch.randint(0, H, (n * bs,), generator=g), torch.randint(0, W, (n * bs,), generator=g)], 1).int()
    lab = torch.randint(0, K, (n * bs,), generator=g)
    gs = torch.amp.GradScaler('cpu', init_scale=2.0 ** 10, growth_interval=2)
    opt = torch.optim.Adam(ref1.parameters(), lr=1e-3)
    want = []
    for s in range(n):
        data = materialise4(scenes, xy[s * bs:(s + 1) * bs].numpy(), P)
        opt.zero_grad()
        loss = dref.qua_loss(ref1(data), bs, lab[s * bs:(s + 1) * bs].float(), dqtl['alpha'], dqtl['beta'], dqtl['gamma'],
                             dqtl['epsilon'], dqtl['tao'])
        gs.scale(loss).backward()
        gs.step(opt); gs.update()
        want.append(loss.item())
    sc = LossScaler('cuda:0', init_scale=2.0 ** 10, growth_interval=2)
    eng = QuaTrainEngine(hip1, QuaScene(scenes, 'cuda:0', half=True), bs, dqtl, lr=1e-3, scaler=sc)
    eng.load_plan(xy, lab)
    eng.run_plan(n, steps_per_graph=3)
    assert sc.get_scale() == gs.get_scale() and sc.skipped_steps() == 0
    assert np.allclose(eng.losses().numpy(), want, atol=2e-5, rtol=1e-5)
    sd = ref1.state_dict()
    for k, v in hip1.state_dict().items():
        # (six Adam steps: an element whose gradient is ~ eps moves by up to lr per step whatever its size)
        assert_close(v, sd[k], 1e-4, 1e-4, 'stage-2 param %s (half scene, scaled)' % k)


@pytest.mark.parametrize('kind', ['SGD', 'RMSprop'])
@pytest.mark.parametrize('graph', [0, 3])
def test_engine_with_the_references_other_optimizers(kind, graph):
    """`make_optimizer` also offers SGD(lr, momentum) and RMSprop(lr, alpha) (utils/utils.py:13-16).  The resident-scene
    engine runs them as a third launch on the flat gradient (dmf_sgd_step / dmf_rmsprop_step); trajectory against the
    oracle net driven by torch's own optimiser on the same batches, eager and from a captured graph."""
    from dmf.engine import Scene, TrainEngine
    from test_gpu_parity import nets
    name = 'tiny1'
    C, C2, P, S, K = SHAPES[name]
    cfg, ref, hip = nets(name)
    n, B, H, W = 9, 16, 23, 19
    A, Bm = scene(name, H, W, 31)
    g = torch.Generator().manual_seed(32)
    xy = torch.stack([torch.randint(0, H, (n * B,), generator=g), torch.randint(0, W, (n * B,), generator=g)], 1).int()
    t = torch.randint(0, K, (n * B,), generator=g)
    if kind == 'SGD':
        opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9)
        kw = dict(optimizer='SGD', momentum=0.9, lr=0.05)
    else:
        opt = torch.optim.RMSprop(ref.parameters(), lr=2e-3, alpha=0.9)
        kw = dict(optimizer='RMSprop', alpha=0.9, lr=2e-3)
    want = []
    for s in range(n):
        a, b = cut(A, Bm, xy[s * B:(s + 1) * B], P, S)
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(ref(a, b), t[s * B:(s + 1) * B])
        loss.backward()
        opt.step()
        want.append(loss.item())
    eng = TrainEngine(hip, Scene(A.numpy(), Bm.numpy(), 'cuda:0'), B, **kw)
    eng.load_plan(xy, t)
    eng.run_plan(n, steps_per_graph=graph)
    assert np.allclose(eng.mean_losses().numpy(), want, atol=2e-5), (eng.mean_losses().numpy(), want)
    sd = ref.state_dict()
    for k, v in hip.state_dict().items():
        assert_close(v, sd[k], 3e-5, 2e-4, '%s: param %s after %d steps' % (kind, k, n))


@pytest.mark.parametrize('kind', ['SGD', 'RMSprop'])
def test_stage2_engine_with_the_references_other_optimizers(kind):
    """Stage 2 (qua_loss, unit-gradient step) under SGD / RMSprop against the oracle's loop (oracle/solver_ref.py::
    qua_train_steps) driven by torch's optimiser.

    Two parts.  (1) Free running from a captured graph: the per-step losses follow the oracle.  (2) Step by step from the
    ORACLE's state (weights and optimiser state are copied into the engine before every step): the update of every element
    whose gradient stands clear of rounding noise agrees tightly, the others are bounded by what the optimiser can make of a
    noise-level gradient.  A free-running comparison of the weights proves nothing here: RMSprop turns a gradient of 1e-8
    into a step of ~lr, and the CPU oracle against ITSELF at 1 / 16 threads (a different partition of its reductions) is
    2.2e-3 apart after three of these steps (tools/diag_rms.py) — the same size as its distance to the GPU."""
    from dmf.engine import QuaScene, QuaTrainEngine
    from oracle.solver_ref import qua_train_steps
    from oracle.gmfnet_ref import Net as RefNet
    from model.gmfnet import Net as HipNet, PARAM_ORDER
    C, C2, P, S, K = SHAPES['qua']
    cfg = make_cfg('qua')
    cfg['gmf']['single_input'] = 1
    torch.manual_seed(5)
    ref = RefNet(cfg)
    hip = HipNet(cfg); hip.load_state_dict(ref.state_dict()); hip = hip.cuda()
    dqtl = {'alpha': 1.0, 'beta': 0.5, 'gamma': 0.5, 'epsilon': 1e-8, 'tao': 2.0}
    g = torch.Generator().manual_seed(6)
    H, W, bs, n = 20, 18, 8, 6
    scenes = [(torch.rand(H + P - 1, W + P - 1, C, generator=g) - 0.2).numpy() for _ in range(4)]
    xy = torch.stack([torch.randint(0, H, (n * bs,), generator=g), torch.randint(0, W, (n * bs,), generator=g)], 1).int()
    lab = torch.randint(0, K, (n * bs,), generator=g)
    lr = 0.05 if kind == 'SGD' else 2e-3
    if kind == 'SGD':
        opt, kw = torch.optim.SGD(ref.parameters(), lr=lr, momentum=0.9), dict(optimizer='SGD', lr=lr, momentum=0.9)
    else:
        opt, kw = torch.optim.RMSprop(ref.parameters(), lr=lr, alpha=0.9), dict(optimizer='RMSprop', lr=lr, alpha=0.9)
    qscene = QuaScene(scenes, 'cuda:0')
    eng = QuaTrainEngine(hip, qscene, bs, dqtl, **kw)
    off = hip._offsets
    named = dict(ref.named_parameters())
    key = 'momentum_buffer' if kind == 'SGD' else 'square_avg'
    want = []
    for i in range(n):
        # the oracle's state BEFORE step i -> the engine
        hip.load_state_dict(ref.state_dict())
        eng.theta = hip.flat_parameters()
        for j, k in enumerate(PARAM_ORDER):
            st = opt.state.get(named[k], {})
            eng.m[off[j]:off[j] + named[k].numel()] = (st[key].reshape(-1) if key in st else torch.zeros(named[k].numel())).cuda()
        eng.step_count = i
        eng.dev_step.fill_(i)
        before = {k: v.detach().clone() for k, v in named.items()}
        l, _ = qua_train_steps(ref, scenes, xy.numpy(), lab.numpy(), bs, P, dqtl, optimizer=opt, batches=[np.arange(i * bs, (i + 1) * bs)])
        want += l
        eng.step(xy[i * bs:(i + 1) * bs], lab[i * bs:(i + 1) * bs])
        assert abs(float(eng.loss.item()) - l[0]) < 2e-5, (i, float(eng.loss.item()), l[0])
        got = dict(hip.named_parameters())
        for k, p_ in named.items():
            gr = p_.grad.detach().abs()
            solid = gr > 1e-3 * gr.max().clamp_min(1e-30)
            d_ref, d_got = p_.detach() - before[k], got[k].detach().cpu() - before[k]
            err = (d_got - d_ref).abs()
            assert (err[solid] <= 5e-6 + 5e-3 * d_ref.abs()[solid]).all(), (kind, i, k, float(err[solid].max()))   # (updates are ~lr .. 3 lr)
            assert float(err.max()) <= 3.3 * lr, (kind, i, k, float(err.max()))       # a noise-level gradient: at most ~3.2 lr
    # free running, from a captured graph: the losses of the first steps (before the weights can have drifted apart)
    torch.manual_seed(5)
    ref2 = RefNet(cfg)
    hip2 = HipNet(cfg); hip2.load_state_dict(ref2.state_dict()); hip2 = hip2.cuda()
    eng2 = QuaTrainEngine(hip2, qscene, bs, dqtl, **kw)
    eng2.load_plan(xy, lab)
    eng2.run_plan(n, steps_per_graph=3)
    opt2 = (torch.optim.SGD(ref2.parameters(), lr=lr, momentum=0.9) if kind == 'SGD' else torch.optim.RMSprop(ref2.parameters(), lr=lr, alpha=0.9))
    want2, _ = qua_train_steps(ref2, scenes, xy.numpy(), lab.numpy(), bs, P, dqtl, optimizer=opt2)
    assert np.allclose(eng2.losses().numpy(), want2, atol=1e-4, rtol=1e-4), (eng2.losses().numpy(), want2)


@pytest.mark.parametrize('half', [False, True])
def test_warm_graph_leaves_no_trace(half):
    """TrainEngine.warm_graph() replays the captured graph once and puts weights, optimiser state, cursors, loss history
    (and the loss scaler's state) back: bench.py uses it when the requested warm-up is shorter than one graph.  A run with
    it must be BIT-identical to a run without it."""
    from dmf.engine import LossScaler, Scene, TrainEngine
    from test_gpu_parity import nets
    name = 'tiny1'
    C, C2, P, S, K = SHAPES[name]
    n, B, H, W, spg = 8, 16, 23, 19, 3
    A, Bm = scene(name, H, W, 41)
    g = torch.Generator().manual_seed(42)
    xy = torch.stack([torch.randint(0, H, (n * B,), generator=g), torch.randint(0, W, (n * B,), generator=g)], 1).int()
    t = torch.randint(0, K, (n * B,), generator=g)
    out = []
    for warm in (False, True):
        cfg, ref, hip = nets(name)
        sc = LossScaler('cuda:0', init_scale=2.0 ** 10, growth_interval=2) if half else None
        eng = TrainEngine(hip, Scene(A.numpy(), Bm.numpy(), 'cuda:0', half=half), B, lr=1e-3, scaler=sc)
        eng.load_plan(xy, t)
        eng.run_plan(2, 0)                              # a warm-up shorter than one graph: eager steps
        eng._capture(spg)
        if warm:
            assert eng.warm_graph()
        eng.run_plan(n - 2, steps_per_graph=spg)        # two replays
        out.append((eng.mean_losses().clone(), eng.theta.clone(), eng.m.clone(), eng.v.clone(),
                    sc.get_scale() if half else None))
    assert out[0][4] == out[1][4]
    for a, b in zip(out[0][:4], out[1][:4]):
        assert torch.equal(a, b)
    assert out[0][0].numel() == n
