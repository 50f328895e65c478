"""Resident-scene training / evaluation engine (the fast path behind the solvers).

What it removes from the reference's hot loop (solver/mainsolver.py:49-58):
  * per-item host patch slicing + 3 H2D copies per step  -> padded scenes live in HBM, patches are
    gathered on-device from pixel coordinates (`dmf_input` mode 1);
  * forward / CE / backward / Adam as separate framework ops -> two launches per step
    (`dmf_train_fwd_bwd`, `dmf_grad_reduce_adam`; with `gmf.attention` four: token kernel, attention fwd+bwd kernel,
    conv backward, reduce+Adam — `dmf_train_attn_fwd_bwd`);
  * `loss.item()` every step                               -> per-step mean loss kept on the device;
  * per-launch host work                                   -> an epoch plan (shuffled coordinates + labels) is
    uploaded once and a captured hipGraph of `steps_per_graph` steps is replayed; batch cursor and the
    Adam step count live in device memory.
Data parallel (world_size > 1), two forms with identical updates on every rank:
  * with a `dmf.xgmi.Communicator`: `dmf_grad_reduce_xgmi_adam` — the ranks exchange their flat gradient inside the
    reduce + Adam launch (one-shot over xGMI, rank-ordered sum), still two launches per step and graph-replayable;
  * without: `dmf_grad_reduce` -> all-reduce(sum) of ONE flat fp32 gradient over RCCL -> `dmf_adam_step(1/world)`.
"""
import numpy as np
import torch

from . import lib


_hip_graph_upload = None


def _upload_graph(g):
    """hipGraphUpload of a freshly captured graph: its first replay otherwise pays the upload (~20 us, measured with
    tools/graph_first.py) inside whatever the caller is timing.  Returns True when the upload happened; a missing symbol or a
    non-zero return code leaves the lazy upload in place (correct either way, only the first replay is slower)."""
    global _hip_graph_upload
    import ctypes
    if _hip_graph_upload is None:
        rt = lib.hip_runtime_of_torch()
        fn = getattr(rt, 'hipGraphUpload', None) if rt is not None else None
        if fn is None:
            _hip_graph_upload = False
        else:
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            fn.restype = ctypes.c_int
            _hip_graph_upload = fn
    if not _hip_graph_upload:
        return False
    rc = _hip_graph_upload(ctypes.c_void_p(g.raw_cuda_graph_exec()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return rc == 0


class Scene:
    """Padded, normalised scenes resident in HBM: A [Hp, Wp, C], B [HpB, WpB, C2] (pixel-major, fp32).
    half: A is kept as IEEE fp16 (`gmf.half`; numpy's float32 -> float16 rounds to nearest even, as the oracle does)."""

    def __init__(self, primary, aux, device, half=False):
        A = np.ascontiguousarray(primary, dtype=np.float32)
        Bm = np.ascontiguousarray(aux, dtype=np.float32)
        if Bm.ndim == 2:
            Bm = Bm[:, :, None]
        self.half = bool(half)
        self.A = torch.from_numpy(A.astype(np.float16) if half else A).to(device)
        self.B = torch.from_numpy(Bm).to(device)
        self.device = torch.device(device)


class LossScaler:
    """Device-resident dynamic loss scale — the role of `torch.cuda.amp.GradScaler` (tostagesolver.py:83-84 builds two
    with torch's defaults; :98 / :119 scale -> step -> update).  Same defaults and update rule; the state never visits
    the host, so a captured graph carries it."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = torch.zeros(lib.SCALER_FLOATS, device=device)
        lib.scaler_init(self.state, float(init_scale))

    def get_scale(self):
        return float(self.state[0].item())

    def skipped_steps(self):
        return int(self.state[3].item())

    def hparams(self):
        return (self.growth_factor, self.backoff_factor, self.growth_interval)


class TrainEngine:
    def __init__(self, net, scene, batch, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, comm=None, scaler=None,
                 optimizer='ADAM', momentum=0.0, alpha=0.99):
        """optimizer: 'ADAM' (fused into the reduce launch), or the reference's other two (utils/utils.py:13-16) — 'SGD'
        (`momentum`) and 'RMSprop' (`alpha`, eps 1e-8) — as a third launch on the flat gradient."""
        if optimizer not in ('ADAM', 'SGD', 'RMSprop'):
            raise lib.DmfError('optimizer %r is not one of ADAM, SGD, RMSprop' % (optimizer,))
        self.optim, self.momentum, self.alpha = optimizer, float(momentum), float(alpha)
        self.net, self.scene, self.B = net, scene, int(batch)
        self.shape = net.shape
        lib.shape_supported(self.shape)
        if getattr(scene, 'half', False):
            lib.require_half(self.shape)
        self.scaler = scaler
        if scaler is not None and (comm is not None or self.shape.attention):
            raise lib.DmfError('loss scaling: late-fusion net, single GPU or RCCL data parallel (not the xgmi exchange)')
        if optimizer != 'ADAM' and (comm is not None or scaler is not None):
            raise lib.DmfError('%s: single GPU or RCCL data parallel, no loss scaler (the fused exchange and the scaler step are ADAM)' % optimizer)
        self.lr, self.b1, self.b2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        dev = scene.device
        self.theta = net.flat_parameters()
        if self.theta.device != dev:
            raise lib.DmfError('net and scene must be on the same device')
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.grad = torch.zeros_like(self.theta)
        K = net.arch['K']
        self.logits = torch.empty(self.B, K, device=dev)
        self.loss = torch.zeros(self.B, device=dev)
        self.ws = torch.empty(lib.workspace_bytes(self.shape, self.B) // 4, device=dev)
        self.attn_ws = None
        if self.shape.attention:                      # token maps + dense gradient maps of the attention block
            self.attn_ws = torch.empty(lib.attn_train_workspace_bytes(self.shape, self.B), dtype=torch.uint8, device=dev)
        self.step_count = 0
        self.pg = process_group
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        self.comm = comm if self.world > 1 else None
        if self.comm is not None and (self.comm.world != self.world or self.comm.capacity < self.theta.numel()):
            raise lib.DmfError('xgmi communicator does not match this engine (world / capacity)')
        # device-side bookkeeping for graph replay
        self.dev_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dev_cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.plan_xy = self.plan_labels = self.plan_pack = self.loss_hist = None
        self.host_cursor = 0
        self.graph = None
        self.graph_steps = 0
        self.graph_hparams = None

    # ------------------------------------------------------------------ eager step (host-side step count)
    def step(self, xy, labels, check=True):
        """One optimiser step on the patches at `xy` [B,2] int32 (device) with `labels` [B] int32 (device)."""
        if xy.shape[0] > self.B:
            raise lib.DmfError('engine was built for batches of at most %d, got %d' % (self.B, xy.shape[0]))
        if check:     # one D2H copy per call; load_plan() validates a whole epoch at once and run_plan() skips this
            lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xy.cpu().numpy())
            if labels.numel() and (int(labels.min()) < 0 or int(labels.max()) >= self.net.arch['K']):
                raise lib.DmfError('label outside [0, %d)' % self.net.arch['K'])
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, xy)
        self._launch(inp, labels, None, None)

    def step_patches(self, a, b, labels):
        """Same step from materialised patch tensors (the reference dataloader's batch)."""
        inp = lib.input_patches(self.shape, a, b)
        self._launch(inp, labels, None, None)

    def _launch(self, inp, labels, dev_step, dev_cursor, loss_hist=None):
        self.step_count += 1
        theta = self.theta
        nB = inp.B
        if (self.comm is not None or self.scaler is not None) and dev_step is None:
            dev_step = self.dev_step                     # the exchange numbers its rounds by the device step count; with a
                                                         # loss scaler skipped steps make the device count the only true one
        if self.optim != 'ADAM':
            # forward + loss + backward, flat gradient [all-reduce], then the optimiser's own launch; m holds its one state
            # vector (SGD: momentum buffer, RMSprop: running mean of squares)
            if dev_step is None:
                dev_step = self.dev_step                 # (SGD's first step is told by the device count: the patch kernel advances it)
            if self.shape.attention:
                lib.train_attn_fwd_bwd(self.shape, inp, theta, self.net.pool_w, labels, None, 1.0 / nB, self.logits, self.loss,
                                       self.ws, self.attn_ws, adam_step_dev=dev_step)
            else:
                lib.train_fwd_bwd(self.shape, inp, theta, self.net.pool_w, labels, 1.0 / nB, self.logits, self.loss, self.ws,
                                  adam_step_dev=dev_step)
            lib.grad_reduce(self.shape, nB, self.ws, self.grad)
            if not self._single():
                import torch.distributed as dist
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.pg)
            if loss_hist is not None:
                loss_hist.scatter_(0, dev_cursor.long(), self.loss[:nB].mean().reshape(1))
            if self.optim == 'SGD':
                lib.sgd_step(theta, self.grad, self.m, self.lr, self.momentum, self.step_count, grad_scale=1.0 / self.world,
                             step_dev=dev_step, cursor_dev=dev_cursor)
            else:
                lib.rmsprop_step(theta, self.grad, self.m, self.lr, self.alpha, grad_scale=1.0 / self.world, cursor_dev=dev_cursor)
            return
        if self.scaler is not None:
            # scaler.scale(loss).backward() -> [all-reduce] -> scaler.unscale_ + scaler.step(opt) + scaler.update()
            sc = self.scaler
            lib.train_fwd_bwd(self.shape, inp, theta, self.net.pool_w, labels, 1.0 / nB, self.logits, self.loss, self.ws,
                              adam_step_dev=dev_step, scaler_state=sc.state)
            if self._single():                          # three launches: patch kernel, reduce (+ unscale + check), Adam-or-skip
                lib.grad_reduce_scaled(self.shape, nB, self.ws, self.grad, sc.state, cursor_dev=dev_cursor,
                                       loss=self.loss if loss_hist is not None else None, loss_hist=loss_hist)
                lib.unscale_adam(theta, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, sc.state,
                                 sc.growth_factor, sc.backoff_factor, sc.growth_interval, dev_step, unscaled=True)
                return
            import torch.distributed as dist
            lib.grad_reduce(self.shape, nB, self.ws, self.grad)
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.pg)     # the check must see the SUM: after it
            if loss_hist is not None:
                loss_hist.scatter_(0, dev_cursor.long(), self.loss[:nB].mean().reshape(1))
            lib.unscale_adam(theta, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, sc.state,
                             sc.growth_factor, sc.backoff_factor, sc.growth_interval, dev_step,
                             grad_scale=1.0 / self.world, cursor_dev=dev_cursor)
            return
        if self.shape.attention:
            lib.train_attn_fwd_bwd(self.shape, inp, theta, self.net.pool_w, labels, None, 1.0 / nB, self.logits, self.loss,
                                   self.ws, self.attn_ws, adam_step_dev=dev_step)
        else:
            lib.train_fwd_bwd(self.shape, inp, theta, self.net.pool_w, labels, 1.0 / nB, self.logits, self.loss, self.ws,
                              adam_step_dev=dev_step)
        if self._single():
            lib.grad_reduce_adam(self.shape, nB, self.ws, theta, self.m, self.v, None, self.lr, self.b1, self.b2, self.eps,
                                 self.step_count, adam_step_dev=dev_step, cursor_dev=dev_cursor,
                                 loss=self.loss if loss_hist is not None else None, loss_hist=loss_hist)
        elif self.comm is not None:
            lib.grad_reduce_xgmi_adam(self.shape, nB, self.ws, theta, self.m, self.v, self.comm.c, self.lr, self.b1,
                                      self.b2, self.eps, 1.0 / self.world, dev_step, cursor_dev=dev_cursor,
                                      loss=self.loss if loss_hist is not None else None, loss_hist=loss_hist)
        else:
            import torch.distributed as dist
            lib.grad_reduce(self.shape, nB, self.ws, self.grad)
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.pg)
            if loss_hist is not None:                    # (this rank's mean loss of the step, as the fused forms record it)
                loss_hist.scatter_(0, dev_cursor.long(), self.loss[:nB].mean().reshape(1))
            lib.adam_step(theta, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, self.step_count,
                          grad_scale=1.0 / self.world, adam_step_dev=dev_step, cursor_dev=dev_cursor)

    # ------------------------------------------------------------------ epoch plan + hipGraph replay
    def load_plan(self, xy_all, labels_all):
        """Upload an epoch's shuffled stream: xy_all [n*B, 2], labels_all [n*B] (host or device, any int type)."""
        dev = self.scene.device
        xy = torch.as_tensor(xy_all).to(device=dev, dtype=torch.int32).contiguous()
        lab = torch.as_tensor(labels_all).to(device=dev, dtype=torch.int32).contiguous()
        if xy.shape[0] % self.B or xy.shape[0] != lab.shape[0]:
            raise lib.DmfError('plan length must be a multiple of the batch size')
        xy_host = xy.cpu().numpy()
        lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xy_host)
        K = self.net.arch['K']
        if int(lab.min()) < 0 or int(lab.max()) >= K:
            raise lib.DmfError('label outside [0, %d)' % K)
        n = xy.shape[0] // self.B
        same = self.plan_xy is not None and self.plan_xy.shape == xy.shape
        # the same stream once more, step by step [n][2B coordinates | B labels]: the window of a captured graph is refilled
        # from it with ONE device copy per replay
        pack = torch.cat([xy.view(n, 2 * self.B), lab.view(n, self.B)], 1).contiguous()
        if same:                                   # keep addresses stable for an already captured graph
            self.plan_xy.copy_(xy); self.plan_labels.copy_(lab); self.plan_pack.copy_(pack)
        else:
            self.plan_xy, self.plan_labels, self.plan_pack = xy, lab, pack
            self.loss_hist = torch.zeros(n, device=dev)
            self.graph = None
        self.dev_cursor.zero_()
        self.host_cursor = 0
        if self.scaler is None and self.optim == 'ADAM':   # (with a scaler the device count is authoritative: skipped steps;
            self.dev_step.fill_(self.step_count)           #  the other optimisers always step by the device count)
        self.plan_steps = n
        return n

    def _plan_step(self):
        if self.host_cursor >= self.plan_steps:          # the kernel reads plan[cursor] unchecked: never step past the plan
            raise lib.DmfError('the loaded plan has %d steps, all of them are done' % self.plan_steps)
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, self.plan_xy, B=self.B, cursor=self.dev_cursor)
        self._launch(inp, self.plan_labels, self.dev_step, self.dev_cursor, self.loss_hist)
        self.host_cursor += 1

    def _fill_window(self, n):
        """Copy the next n steps of the plan into the fixed window the captured graph reads (one small async copy)."""
        self.win.copy_(self.plan_pack[self.host_cursor:self.host_cursor + n])

    def run_plan(self, steps=None, steps_per_graph=0):
        """Run `steps` steps of the loaded plan (default: all).  steps_per_graph > 0 replays a captured hipGraph
        (single GPU, or data parallel over the xgmi communicator); 0 launches eagerly.  No host synchronisation."""
        steps = self.plan_steps - self.host_cursor if steps is None else steps
        if self.plan_xy is None or steps < 0 or self.host_cursor + steps > self.plan_steps:
            raise lib.DmfError('run_plan(%d): the loaded plan has %d steps, %d of them done' % (
                steps, getattr(self, 'plan_steps', 0), self.host_cursor))
        done = 0
        if steps_per_graph < 0 and self._native_loop_ok():
            # the library's own loop: 2 launches per step enqueued from C, the batches read straight from the plan (no window,
            # no graph): for a short run the fixed cost is one kernel launch instead of a window copy + a graph launch
            k0 = self.host_cursor
            inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, self.plan_xy[k0 * self.B:(k0 + steps) * self.B], B=self.B)
            lib.train_plan_steps(self.shape, inp, self.theta, self.net.pool_w, self.plan_labels[k0 * self.B:(k0 + steps) * self.B],
                                 1.0 / self.B, self.logits, self.loss, self.ws, self.m, self.v, self.lr, self.b1, self.b2, self.eps,
                                 self.dev_step, self.dev_cursor, self.loss_hist, steps)
            self.step_count += steps
            self.host_cursor += steps
            return steps
        if steps_per_graph > 0 and self._graphable():
            # lr, betas and eps are launch arguments baked into the captured graph (reference: `scheduler.step()` changes
            # the optimiser's lr every epoch, mainsolver.py:60): a change invalidates the graph
            if self.graph is None or self.graph_steps != steps_per_graph or self.graph_hparams != self._hparams():
                try:
                    self._capture(steps_per_graph)
                except RuntimeError:
                    if self._single() or self.comm is not None:
                        raise
                    # RCCL's all-reduce refused to be captured (every rank runs the same software, so every rank lands
                    # here): stay on eager launches for the rest of this engine's life
                    self._rccl_graph = False
                    self.graph = None
                    torch.cuda.synchronize()
            while self.graph is not None and steps - done >= steps_per_graph:
                self._fill_window(steps_per_graph)
                self.graph.replay()
                self.step_count += steps_per_graph
                self.host_cursor += steps_per_graph
                done += steps_per_graph
        for _ in range(steps - done):
            self._plan_step()
        return steps

    def _single(self):
        """One rank and no collective in the step.  `_force_collective` (tests only) keeps the group's all-reduce in the step of
        a ONE-rank group, so that the RCCL form of the step — eager and captured in a hipGraph — runs on a one-GPU box."""
        return self.world == 1 and not getattr(self, '_force_collective', False)

    def _native_loop_ok(self):
        """run_plan(steps, steps_per_graph=-1): the C loop of dmf_train_plan_steps — late-fusion net, ADAM, one GPU, no scaler."""
        return self._single() and self.scaler is None and self.optim == 'ADAM' and not self.shape.attention

    def _graphable(self):
        """Can a step be captured in a hipGraph?  One GPU: yes.  The one-shot exchange: yes (it is part of the reduce launch).
        The RCCL all-reduce: yes when the process group is RCCL — its collectives are capturable, and a host-enqueued
        all_reduce of 32 KB per ~16-us step would otherwise bound the step by the host (DMF_RCCL_GRAPH=0 switches this off)."""
        if self._single() or self.comm is not None:
            return True
        if self.scaler is not None or self.optim != 'ADAM' or not getattr(self, '_rccl_graph', True):
            return False
        import os
        import torch.distributed as dist
        return dist.get_backend(self.pg) == 'nccl' and os.environ.get('DMF_RCCL_GRAPH', '1') != '0'

    def _capture(self, n):
        # hipFuncSetAttribute is not capturable, so every kernel must have been launched once before the capture:
        # run one step eagerly, then put back the exact pre-step state (capture itself executes nothing).
        state = (self.theta, self.m, self.v, self.dev_step, self.dev_cursor, self.loss_hist)
        if self.scaler is not None:
            state = state + (self.scaler.state,)
        count0 = self.step_count
        saved = [t.clone() for t in state]
        self._plan_step()
        torch.cuda.synchronize()
        for t, s in zip(state, saved):
            t.copy_(s)
        self.step_count = count0
        if self.comm is not None:                  # the eager step used up an exchange sequence number; the bias is
            self.comm.rewind(1)                    # a launch argument, so it has to move BEFORE the capture
        self.host_cursor -= 1                      # (the eager step above advanced it)
        # Inside the graph step k reads its coordinates and labels from slot k of a FIXED window (plain pointers baked
        # into the launch) instead of plan[cursor]: every kernel starts cold after the previous kernel's cache
        # write-back / invalidate, and `kernarg -> cursor -> coordinates -> gather` is one dependent miss longer than
        # `kernarg -> coordinates -> gather`.  The window is refilled from the plan before every replay.
        dev = self.scene.device
        self.win = torch.empty(n, 3 * self.B, dtype=torch.int32, device=dev)      # per step: 2B coordinates, B labels
        if self.host_cursor + n <= self.plan_steps:
            self._fill_window(n)
        else:
            self.win.zero_()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(n):
                inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, self.win[k, :2 * self.B].view(self.B, 2))
                self._launch(inp, self.win[k, 2 * self.B:], self.dev_step, self.dev_cursor, self.loss_hist)
        self.step_count = count0
        self.graph, self.graph_steps, self.graph_hparams = g, n, self._hparams()
        _upload_graph(g)

    def warm_graph(self):
        """One replay of the captured graph that leaves no trace (weights, optimiser state, cursors and loss history are put
        back): the first launch of a graph executable pays one-time costs of the launch machinery, which belong to a
        warm-up.  bench.py calls it when the requested warm-up is shorter than one graph.  Single GPU only: under the
        xGMI exchange a replay also advances the sequence numbers in the peers' inboxes, which cannot be put back."""
        if self.graph is None or not self._single() or self.host_cursor + self.graph_steps > self.plan_steps:
            return False
        state = (self.theta, self.m, self.v, self.dev_step, self.dev_cursor, self.loss_hist)
        if self.scaler is not None:
            state = state + (self.scaler.state,)
        saved = [t.clone() for t in state]
        self._fill_window(self.graph_steps)
        self.graph.replay()
        torch.cuda.synchronize()
        for t, s_ in zip(state, saved):
            t.copy_(s_)
        torch.cuda.synchronize()
        return True

    def _hparams(self):
        return (self.lr, self.b1, self.b2, self.eps, self.momentum, self.alpha) + (self.scaler.hparams() if self.scaler is not None else ())

    def mean_losses(self):
        """Per-step mean CE of the plan steps run so far (one D2H copy)."""
        return self.loss_hist[:int(self.dev_cursor.item())].cpu()


class EvalEngine:
    """Forward + argmax + on-device confusion matrix / label map (mainsolver.py:102-147,164-197)."""

    def __init__(self, net, scene, batch):
        self.net, self.scene, self.B = net, scene, int(batch)
        self.shape = net.shape
        lib.shape_supported(self.shape)
        dev = scene.device
        K = net.arch['K']
        self.logits = torch.empty(self.B, K, device=dev)
        self.pred = torch.empty(self.B, dtype=torch.int32, device=dev)
        self.attn_ws = None
        if self.shape.attention:
            self.attn_ws = torch.empty(lib.attn_workspace_bytes(self.shape, self.B), dtype=torch.uint8, device=dev)

    def predict(self, xy):
        """xy [n,2] int32 device -> (logits [n,K], pred [n]) views valid until the next call."""
        n = xy.shape[0]
        if n > self.B:
            raise lib.DmfError('batch larger than the engine was built for')
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, xy)
        if self.shape.attention:
            lib.forward_attn(self.shape, inp, self.net.flat_parameters(), self.net.pool_w, self.attn_ws, self.logits, self.pred)
        else:
            lib.forward(self.shape, inp, self.net.flat_parameters(), self.net.pool_w, self.logits, self.pred)
        return self.logits[:n], self.pred[:n]

    def ce_sum(self, xy, labels):
        """Sum over the batch of the per-patch cross-entropy (device scalar, float64), from the evaluation launch itself
        (dmf_forward_ce); None where the shape has no such kernel."""
        n = xy.shape[0]
        if n > self.B:
            raise lib.DmfError('batch larger than the engine was built for')
        if self.shape.attention or getattr(self, '_no_ce', False):
            return None
        if not hasattr(self, 'ce'):
            self.ce = torch.zeros(self.B, device=self.scene.device)
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, xy)
        try:
            lib.forward_ce(self.shape, inp, self.net.flat_parameters(), self.net.pool_w, labels, self.logits, self.ce, self.pred)
        except lib.DmfError:
            self._no_ce = True
            return None
        return self.ce[:n].double().sum()

    def confusion(self, xy_all, labels_all, matrix=None, process_group=None):
        """Confusion matrix [K,K] int64 (rows = prediction) over all given pixels; one D2H at the end.
        With a process group every rank classifies its contiguous shard of the pixels and the matrices are summed."""
        dev = self.scene.device
        K = self.net.arch['K']
        xy_all = torch.as_tensor(xy_all).to(device=dev, dtype=torch.int32).contiguous()
        labels_all = torch.as_tensor(labels_all).to(device=dev, dtype=torch.int32).contiguous()
        if process_group is not None:
            import torch.distributed as dist
            from .parallel import shard_range
            lo, hi = shard_range(xy_all.shape[0], dist.get_rank(process_group), dist.get_world_size(process_group))
            xy_all, labels_all = xy_all[lo:hi].contiguous(), labels_all[lo:hi].contiguous()
        lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xy_all.cpu().numpy())
        if matrix is None:
            matrix = torch.zeros(K, K, dtype=torch.int64, device=dev)
        for i in range(0, xy_all.shape[0], self.B):
            xy = xy_all[i:i + self.B]
            _, pred = self.predict(xy)
            lib.confusion_accum(pred, labels_all[i:i + self.B], K, matrix)
        if process_group is not None:
            from .parallel import allreduce_sum_
            if dist.get_backend(process_group) == 'nccl':
                allreduce_sum_(matrix, process_group)
            else:                                   # gloo (CPU tests, one-GPU rehearsal): reduce on the host
                matrix.copy_(allreduce_sum_(matrix.cpu(), process_group))
        return matrix

    def label_map(self, xy_all, H, W, label_map=None, process_group=None):
        """Predicted class of every given pixel written at (x, y) of an [H, W] int32 map (mainsolver.py:171-183).
        With a process group the pixels are sharded and the tiles merged (every pixel is written by one rank)."""
        dev = self.scene.device
        xy_all = torch.as_tensor(xy_all).to(device=dev, dtype=torch.int32).contiguous()
        if process_group is not None:
            import torch.distributed as dist
            from .parallel import shard_range
            lo, hi = shard_range(xy_all.shape[0], dist.get_rank(process_group), dist.get_world_size(process_group))
            xy_all = xy_all[lo:hi].contiguous()
        xy_host = xy_all.cpu().numpy()
        lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xy_host)
        if len(xy_host) and (int(xy_host[:, 0].max()) >= H or int(xy_host[:, 1].max()) >= W):
            raise lib.DmfError('pixel outside the %d x %d label map' % (H, W))      # labelmap_kernel writes map[x * W + y]
        if label_map is None:
            label_map = torch.zeros(H, W, dtype=torch.int32, device=dev)
        for i in range(0, xy_all.shape[0], self.B):
            xy = xy_all[i:i + self.B]
            _, pred = self.predict(xy)
            lib.labelmap_write(pred, xy, W, label_map)
        if process_group is not None:
            from .parallel import allreduce_max_
            if dist.get_backend(process_group) == 'nccl':
                allreduce_max_(label_map, process_group)
            else:
                label_map.copy_(allreduce_max_(label_map.cpu(), process_group))
        return label_map


# ====================================================================== stage 2 of the two-stage path
class QuaScene:
    """The four co-registered padded scenes of stage 2 (ms, pan, ms_gan, pan_gan; tostagesolver.py:248-257) resident
    in HBM as ONE tall pixel-major scene [4*Hp, Wp, C] plus its band mean [4*Hp, Wp, 1] (the single-input net's
    auxiliary modality).  Stream k's patch at pixel (x, y) is the tall scene's patch at (x + k*Hp, y); a window
    never crosses into the next stream because every stream carries its own bottom padding."""

    def __init__(self, scenes, device, half=False):
        if len(scenes) != 4 or any(s.shape != scenes[0].shape for s in scenes):
            raise lib.DmfError('stage 2 wants four scenes of one shape')
        self.Hp = int(scenes[0].shape[0])
        tall = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.float32) for s in scenes], axis=0))
        self.A = torch.from_numpy(tall).to(device)
        self.B = lib.band_mean_scene(self.A)          # (of the fp32 bands: the aux modality stays fp32)
        self.half = bool(half)
        if half:                                      # `gmf.half`: the primary scene is kept as fp16 (round to nearest even)
            self.A = self.A.to(torch.float16)
        self.device = torch.device(device)

    def stack_xy(self, xy, streams=4):
        """[n, 2] pixel coordinates -> [streams*n, 2] coordinates in the tall scene, stream-major like
        `torch.concat([data1, data2, data3, data4])` (tostagesolver.py:272)."""
        xy = torch.as_tensor(xy).to(torch.int32)
        off = torch.zeros_like(xy)
        out = []
        for k in range(streams):
            off[:, 0] = k * self.Hp
            out.append(xy + off)
        return torch.cat(out)


class QuaTrainEngine:
    """Stage-2 train step (tostagesolver.py:268-278) on the resident tall scene, no host sync.  The loss couples the whole
    batch, so it cannot ride inside the per-patch kernel like cross-entropy does.  Two forms:
      * unit-gradient step (shapes with a v2 kernel): `dmf_forward_unit` (forward of the 4*bs stacked patches + the conv
        backward for a unit gradient per pooled feature) -> `dmf_qua_loss` (value + d/dlogits) -> `dmf_backward_unit`
        (dh, dz, scaled slab rows) -> `dmf_grad_reduce_adam`: the patches are visited ONCE, and the four launches replay
        from a captured hipGraph (`run_plan(steps, steps_per_graph)`);
      * otherwise `dmf_forward` -> `dmf_qua_loss` -> `dmf_backward_dlogits` (recomputes the forward) -> reduce + ADAM.
    Data parallel (process_group): every rank takes its shard of each batch; the logits of all ranks are gathered so that
    the batch-coupled loss is the GLOBAL batch's, the flat gradient is all-reduced (sum) and ADAM runs with 1/world."""

    def __init__(self, net, scene, bs, dqtl, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, scaler=None,
                 optimizer='ADAM', momentum=0.0, alpha=0.99):
        if optimizer not in ('ADAM', 'SGD', 'RMSprop'):
            raise lib.DmfError('optimizer %r is not one of ADAM, SGD, RMSprop' % (optimizer,))
        if optimizer != 'ADAM' and scaler is not None:
            raise lib.DmfError('the loss-scaler step is ADAM')
        self.optim, self.momentum, self.alpha = optimizer, float(momentum), float(alpha)
        if not net.arch.get('single_input'):
            raise lib.DmfError('stage 2 needs the single-input net (cfg["gmf"]["single_input"] = 1)')
        self.net, self.scene, self.bs = net, scene, int(bs)
        self.shape = net.shape
        lib.shape_supported(self.shape)
        self.unit = lib.unit_supported(self.shape)
        if getattr(scene, 'half', False):
            lib.require_half(self.shape)
        self.scaler = scaler
        if scaler is not None and (process_group is not None or not self.unit):
            raise lib.DmfError('loss scaling in stage 2: unit-gradient step on one GPU')
        self.params = lib.qua_params(dqtl)
        self.lr, self.b1, self.b2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        dev = scene.device
        self.theta = net.flat_parameters()
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.grad = torch.zeros_like(self.theta)
        K = net.arch['K']
        self.pg, self.world, self.rank = process_group, 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.logits = torch.empty(4 * self.bs, K, device=dev)
        self.dlogits = torch.empty(4 * self.bs, K, device=dev)
        self.loss = torch.zeros(1, device=dev)
        self.ws = torch.empty(lib.workspace_bytes(self.shape, 4 * self.bs) // 4, device=dev)
        self.dev_cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dev_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.step_count = 0
        self.plan_xy = self.plan_labels = self.plan_labels_global = self.loss_hist = None
        self.graph, self.graph_steps, self.graph_hparams = None, 0, None

    def _hparams(self):
        return (self.lr, self.b1, self.b2, self.eps, self.momentum, self.alpha)

    def _optimizer_step(self, theta, grad_scale, dev_step, cursor):
        """SGD / RMSprop on the flat gradient (utils/utils.py:13-16); m holds the optimiser's one state vector."""
        if self.optim == 'SGD':
            lib.sgd_step(theta, self.grad, self.m, self.lr, self.momentum, self.step_count, grad_scale=grad_scale,
                         step_dev=dev_step, cursor_dev=cursor)
        else:
            lib.rmsprop_step(theta, self.grad, self.m, self.lr, self.alpha, grad_scale=grad_scale, cursor_dev=cursor)

    def _step(self, inp, bs, labels, cursor, loss_hist, dev_step=None):
        self.step_count += 1
        theta = self.theta
        if self.scaler is not None:
            # scaler.scale(loss).backward(); scaler.step(opt); scaler.update()  (tostagesolver.py:98,119 do this for the
            # stage-1 pair; here it wraps the stage-2 step)
            sc = self.scaler
            dev_step = self.dev_step if dev_step is None else dev_step
            lib.forward_unit(self.shape, inp, theta, self.net.pool_w, self.logits, self.ws, adam_step_dev=dev_step)
            lib.qua_loss(self.logits[:4 * bs], bs, labels, self.params, loss=self.loss, dlogits=self.dlogits[:4 * bs],
                         cursor=cursor, loss_hist=loss_hist, scaler_state=sc.state)
            lib.backward_unit(self.shape, 4 * bs, theta, self.dlogits, self.ws)
            lib.grad_reduce_scaled(self.shape, 4 * bs, self.ws, self.grad, sc.state, cursor_dev=cursor)
            lib.unscale_adam(theta, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, sc.state,
                             sc.growth_factor, sc.backoff_factor, sc.growth_interval, dev_step, unscaled=True)
            return
        if self.unit:
            lib.forward_unit(self.shape, inp, theta, self.net.pool_w, self.logits, self.ws, adam_step_dev=dev_step)
        else:
            lib.forward(self.shape, inp, theta, self.net.pool_w, self.logits)
        if self.world == 1:
            lib.qua_loss(self.logits[:4 * bs], bs, labels, self.params, loss=self.loss, dlogits=self.dlogits[:4 * bs],
                         cursor=cursor, loss_hist=loss_hist)
        else:
            self._global_loss(bs, labels, cursor, loss_hist)
        if self.unit:
            lib.backward_unit(self.shape, 4 * bs, theta, self.dlogits, self.ws)
        else:
            lib.backward_dlogits(self.shape, inp, theta, self.net.pool_w, self.dlogits, self.ws)
        if self.world == 1 and self.optim != 'ADAM':
            lib.grad_reduce(self.shape, 4 * bs, self.ws, self.grad)
            self._optimizer_step(theta, 1.0, dev_step if self.unit else None, cursor)
        elif self.world == 1:
            lib.grad_reduce_adam(self.shape, 4 * bs, self.ws, theta, self.m, self.v, None, self.lr, self.b1, self.b2, self.eps,
                                 self.step_count, adam_step_dev=dev_step if self.unit else None, cursor_dev=cursor)
        else:
            import torch.distributed as dist
            lib.grad_reduce(self.shape, 4 * bs, self.ws, self.grad)
            if dist.get_backend(self.pg) == 'nccl':
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.pg)
            else:
                g = self.grad.cpu()
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
                self.grad.copy_(g)
            # the loss kernel already divided by the GLOBAL batch (it saw all ranks' logits): the sum over ranks is the gradient
            if self.optim != 'ADAM':
                self._optimizer_step(theta, 1.0, None, cursor)
            else:
                lib.adam_step(theta, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, self.step_count,
                              grad_scale=1.0, cursor_dev=cursor)

    def _global_loss(self, bs, labels, cursor, loss_hist):
        """qua_loss couples all samples of the batch (six batch-mean KL terms): every rank evaluates it on the logits of
        ALL ranks (4*bs*world x K floats, a few tens of KB) and keeps its own rows of d loss / d logits."""
        import torch.distributed as dist
        W, K = self.world, self.logits.shape[1]
        mine = self.logits[:4 * bs].contiguous()
        if dist.get_backend(self.pg) == 'nccl':
            parts = [torch.empty_like(mine) for _ in range(W)]
            dist.all_gather(parts, mine, group=self.pg)
        else:
            host = [torch.empty(4 * bs, K) for _ in range(W)]
            dist.all_gather(host, mine.cpu(), group=self.pg)
            parts = [h.to(mine.device) for h in host]
        # rank r holds [stream][bs] rows; the global batch is [stream][rank][bs]
        glob = torch.stack([p.view(4, bs, K) for p in parts], 1).reshape(4 * W * bs, K).contiguous()
        step = int(self.host_cursor)
        glab = self.plan_labels_global[step * W * bs:(step + 1) * W * bs].contiguous()
        gdl = torch.empty_like(glob)
        lib.qua_loss(glob, W * bs, glab, self.params, loss=self.loss, dlogits=gdl)
        if loss_hist is not None:
            loss_hist[step:step + 1].copy_(self.loss)
        self.dlogits[:4 * bs].copy_(gdl.view(4, W, bs, K)[:, self.rank].reshape(4 * bs, K))

    def step(self, xy, labels):
        """One step on the bs pixels `xy` [bs, 2] (host or device ints) with `labels` [bs]."""
        if self.world > 1:
            raise lib.DmfError('data-parallel stage 2 runs from a plan (load_plan / run_plan)')
        bs = int(xy.shape[0])
        if bs > self.bs:
            raise lib.DmfError('engine was built for batches of at most %d' % self.bs)
        dev = self.scene.device
        xy4 = self.scene.stack_xy(torch.as_tensor(xy).cpu()).to(dev).contiguous()
        lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xy4.cpu().numpy())
        lab = torch.as_tensor(labels).to(device=dev, dtype=torch.int32).contiguous()
        K = self.net.arch['K']
        if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= K):
            raise lib.DmfError('label outside [0, %d)' % K)
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, xy4)
        self._step(inp, bs, lab, None, None)

    def load_plan(self, xy_all, labels_all):
        """An epoch of full GLOBAL batches: xy_all [n*bs*world, 2], labels_all [n*bs*world]; rank r trains on rows
        [r*bs, (r+1)*bs) of every global batch."""
        dev = self.scene.device
        W = self.world
        xy = torch.as_tensor(xy_all).to(torch.int32).cpu()
        lab = torch.as_tensor(labels_all).to(device=dev, dtype=torch.int32).contiguous()
        if xy.shape[0] % (self.bs * W) or xy.shape[0] != lab.shape[0]:
            raise lib.DmfError('plan length must be a multiple of the (global) batch size')
        n = xy.shape[0] // (self.bs * W)
        K = self.net.arch['K']
        if n and (int(lab.min()) < 0 or int(lab.max()) >= K):
            raise lib.DmfError('label outside [0, %d)' % K)
        self.plan_labels_global = lab
        if W > 1:
            xy = xy.view(n, W, self.bs, 2)[:, self.rank].reshape(-1, 2)
            lab = lab.view(n, W, self.bs)[:, self.rank].reshape(-1).contiguous()
        xy4 = torch.cat([self.scene.stack_xy(xy[i * self.bs:(i + 1) * self.bs]) for i in range(n)]) if n else xy
        lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xy4.numpy())
        same = self.plan_xy is not None and self.plan_xy.shape == xy4.shape
        if same:                                   # keep addresses stable for an already captured graph
            self.plan_xy.copy_(xy4); self.plan_labels.copy_(lab)
        else:
            self.plan_xy, self.plan_labels = xy4.to(dev).contiguous(), lab
            self.loss_hist = torch.zeros(max(n, 1), device=dev)
            self.graph = None
        self.loss_hist.zero_()
        self.dev_cursor.zero_()
        self.host_cursor = 0
        if self.scaler is None:
            self.dev_step.fill_(self.step_count)
        self.plan_steps = n
        return n

    def _plan_step(self):
        if self.host_cursor >= self.plan_steps:          # the kernel reads plan[cursor] unchecked: never step past the plan
            raise lib.DmfError('the loaded plan has %d steps, all of them are done' % self.plan_steps)
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, self.plan_xy, B=4 * self.bs, cursor=self.dev_cursor)
        self._step(inp, self.bs, self.plan_labels, self.dev_cursor, self.loss_hist, self.dev_step)
        self.host_cursor += 1

    def _capture(self, n):
        """Capture n steps (unit-gradient form, one GPU).  hipFuncSetAttribute is not capturable: one eager step first,
        then the exact pre-step state is put back (capture itself executes nothing)."""
        if not self.unit or self.world > 1:
            raise lib.DmfError('graph replay needs the unit-gradient step on one GPU')
        state = (self.theta, self.m, self.v, self.dev_step, self.dev_cursor, self.loss_hist)
        if self.scaler is not None:
            state = state + (self.scaler.state,)
        count0 = self.step_count
        saved = [t.clone() for t in state]
        self._plan_step()
        torch.cuda.synchronize()
        for t, s in zip(state, saved):
            t.copy_(s)
        self.step_count = count0
        self.host_cursor -= 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, self.plan_xy, B=4 * self.bs, cursor=self.dev_cursor)
                self._step(inp, self.bs, self.plan_labels, self.dev_cursor, self.loss_hist, self.dev_step)
        self.step_count = count0
        self.graph, self.graph_steps, self.graph_hparams = g, n, self._hparams()
        _upload_graph(g)

    def run_plan(self, steps=None, steps_per_graph=0):
        steps = self.plan_steps - self.host_cursor if steps is None else steps
        if self.plan_xy is None or steps < 0 or self.host_cursor + steps > self.plan_steps:
            raise lib.DmfError('run_plan(%d): the loaded plan has %d steps, %d of them done' % (
                steps, getattr(self, 'plan_steps', 0), self.host_cursor))
        done = 0
        if steps_per_graph > 0 and self.unit and self.world == 1:
            if self.graph is None or self.graph_steps != steps_per_graph or self.graph_hparams != self._hparams():
                self._capture(steps_per_graph)
            while steps - done >= steps_per_graph:
                self.graph.replay()
                self.step_count += steps_per_graph
                self.host_cursor += steps_per_graph
                done += steps_per_graph
        for _ in range(steps - done):
            self._plan_step()
        return steps

    def losses(self):
        return self.loss_hist[:int(self.dev_cursor.item())].cpu()

    # bench.py: the step's dominant launch alone (for HIP-event timing) and its name
    def time_dominant(self, inp):
        if self.unit:
            lib.forward_unit(self.shape, inp, self.theta, self.net.pool_w, self.logits, self.ws)
        else:
            lib.backward_dlogits(self.shape, inp, self.theta, self.net.pool_w, self.dlogits, self.ws)

    def dominant_name(self):
        a = self.net.arch
        if self.unit:
            return 'dmf::patch_v2_kernel<Shape<%d,%d,%d,1,%d,..>, MODE_UNIT%s> (dmf_forward_unit: forward + unit gradients of the 4*bs stacked patches)' % (
                a['C'], a['C2'], a['P'], a['F'], ', fp16 scene' if getattr(self.scene, 'half', False) else '')
        return 'dmf::patch_kernel<ShapeQua, MODE_BWD> (dmf_backward_dlogits: forward recompute + backward of the 4*bs stacked patches)'


class QuaEvalEngine:
    """Stage-2 prediction `(out[:bs] + out[bs:2*bs]).softmax(-1).argmax` (tostagesolver.py:337): only the ms and pan
    streams enter it, so only those two are computed."""

    def __init__(self, net, scene, batch, dqtl=None):
        self.net, self.scene, self.B = net, scene, int(batch)
        self.shape = net.shape
        lib.shape_supported(self.shape)
        dev = scene.device
        self.logits = torch.empty(4 * self.B, net.arch['K'], device=dev)
        self.pred = torch.empty(self.B, dtype=torch.int32, device=dev)
        self.loss = torch.zeros(1, device=dev)
        self.params = lib.qua_params(dqtl) if dqtl is not None else None

    def _forward(self, xy, streams, checked=False):
        n = int(xy.shape[0])
        if n > self.B:
            raise lib.DmfError('batch larger than the engine was built for')
        dev = self.scene.device
        if checked and torch.is_tensor(xy) and xy.device == dev:        # (whole-set passes: bounds checked once, stacking on the device)
            off = torch.zeros(1, 2, dtype=torch.int32, device=dev)
            parts = []
            for k in range(streams):
                off[0, 0] = k * self.scene.Hp
                parts.append(xy + off)
            xyk = torch.cat(parts).contiguous()
        else:
            xyk = self.scene.stack_xy(torch.as_tensor(xy).cpu(), streams).to(dev).contiguous()
            lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, xyk.cpu().numpy())
        inp = lib.input_gather(self.shape, self.scene.A, self.scene.B, xyk)
        lib.forward(self.shape, inp, self.net.flat_parameters(), self.net.pool_w, self.logits)
        return n

    def _checked_all(self, xy_all):
        dev = self.scene.device
        xy_all = torch.as_tensor(xy_all).to(torch.int32)
        host = xy_all.cpu()
        lib.check_xy_bounds(self.shape, self.scene.A, self.scene.B, self.scene.stack_xy(host, 2).numpy())
        return xy_all.to(dev).contiguous(), host.numpy()

    def confusion(self, xy_all, labels_all, matrix=None):
        """Confusion matrix [K,K] int64 (rows = prediction) of the pair prediction over all given pixels, in chunks of the
        engine's size (tostagesolver.py:331-341)."""
        dev = self.scene.device
        K = self.net.arch['K']
        xy_all, _ = self._checked_all(xy_all)
        labels_all = torch.as_tensor(labels_all).to(device=dev, dtype=torch.int32).contiguous()
        if matrix is None:
            matrix = torch.zeros(K, K, dtype=torch.int64, device=dev)
        for i in range(0, xy_all.shape[0], self.B):
            n = self._forward(xy_all[i:i + self.B], 2, checked=True)
            lib.pair_argmax(self.logits, n, self.pred)
            lib.confusion_accum(self.pred[:n], labels_all[i:i + n], K, matrix)
        return matrix

    def label_map(self, xy_all, H, W, label_map=None):
        """Pair prediction of every given pixel written at (x, y) of an [H, W] int32 map (tostagesolver.py:360-383)."""
        dev = self.scene.device
        xy_all, host = self._checked_all(xy_all)
        if len(host) and (int(host[:, 0].max()) >= H or int(host[:, 1].max()) >= W):
            raise lib.DmfError('pixel outside the %d x %d label map' % (H, W))
        if label_map is None:
            label_map = torch.zeros(H, W, dtype=torch.int32, device=dev)
        for i in range(0, xy_all.shape[0], self.B):
            xy = xy_all[i:i + self.B]
            n = self._forward(xy, 2, checked=True)
            lib.pair_argmax(self.logits, n, self.pred)
            lib.labelmap_write(self.pred[:n], xy, W, label_map)
        return label_map

    def predict(self, xy):
        n = self._forward(xy, 2)
        lib.pair_argmax(self.logits, n, self.pred)
        return self.logits[:2 * n], self.pred[:n]

    def loss_value(self, xy, labels):
        """qua_loss of a batch without gradients (the validation loop, tostagesolver.py:288-296); device scalar."""
        n = self._forward(xy, 4)
        lab = torch.as_tensor(labels).to(device=self.scene.device, dtype=torch.int32).contiguous()
        lib.qua_loss(self.logits[:4 * n], n, lab, self.params, loss=self.loss)
        return self.loss
