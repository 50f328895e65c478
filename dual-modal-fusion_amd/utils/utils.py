"""utils.utils — optimiser / loss / scheduler factories and checkpoint helpers (mirror of the reference module).

The three factories return the torch objects the reference builds from the same cfg keys (utils/utils.py:8-71):
`ADAM` is `torch.optim.Adam(params, lr)` with every other default, `Criterion` is `nn.CrossEntropyLoss()`, and the eight
scheduler kinds keep the reference's constants.  Here they are lookup tables of small constructors rather than
if-chains.  They serve the drop-in path (reference-style loop -> `model.gmfnet.Net` -> autograd); the resident-scene
fast path (dmf/engine.py) applies the same ADAM update inside `dmf_grad_reduce_adam` and takes only the hyper-parameters
from here (`adam_hparams`, `epoch_hparams`: whatever scheduler kind is configured).  Checkpoints keep the reference's layout (:82-111): `{'state_dict', 'optimizer'}`.
"""
import os
import random

import numpy as np
import torch
from torch import nn
from torch.optim import lr_scheduler as _sched

_ADAM_DEFAULTS = ((0.9, 0.999), 1e-8)          # betas, eps of torch.optim.Adam — the reference passes lr only (:12)

_OPTIMIZERS = {
    'ADAM': lambda s, params: torch.optim.Adam(params, lr=s['lr']),
    'SGD': lambda s, params: torch.optim.SGD(params, lr=s['lr'], momentum=s['momentum']),
    'RMSprop': lambda s, params: torch.optim.RMSprop(params, lr=s['lr'], alpha=s['alpha']),
}


def _qua():
    from train.loss_function import qua_loss      # imported late: it binds the HIP library
    return qua_loss()


_LOSSES = {
    'MSE': lambda: nn.MSELoss(reduction='mean'),
    'L1': lambda: nn.L1Loss(reduction='mean'),
    'Criterion': nn.CrossEntropyLoss,
    'KL': lambda: nn.KLDivLoss(reduction='batchmean'),
    'qua_loss': _qua,
}


def _warmup(opt):
    return _sched.LinearLR(opt, start_factor=0.1, end_factor=1, total_iters=10)


def _decay(opt):
    return _sched.ExponentialLR(optimizer=opt, gamma=0.98)


# every entry: (optimizer, cfg['schedule'], cfg) -> scheduler; `ratio` = base_lr / lr as the reference forms it
_SCHEDULERS = {
    'StepLR': lambda o, s, c: _sched.StepLR(o, step_size=50, gamma=s['base_lr'] / s['lr']),
    'LinearLR': lambda o, s, c: _warmup(o),
    'CosineAnnealingLR': lambda o, s, c: _sched.CosineAnnealingLR(o, 50, s['base_lr']),
    'CyclicLR': lambda o, s, c: _sched.CyclicLR(o, base_lr=s['base_lr'], max_lr=s['lr'], step_size_up=10,
                                                step_size_down=40, cycle_momentum=False),
    'OneCycleLR': lambda o, s, c: _sched.OneCycleLR(o, max_lr=s['lr'], pct_start=0.5, total_steps=c['epoch'],
                                                    div_factor=s['lr'] / s['base_lr'],
                                                    final_div_factor=s['lr'] / s['base_lr']),
    'ConstantLR': lambda o, s, c: _sched.ConstantLR(o, factor=s['base_lr'] / s['lr'], total_iters=10),
    'ChainedScheduler': lambda o, s, c: _sched.ChainedScheduler([_warmup(o), _decay(o)]),
    'ExponentialLR': lambda o, s, c: _decay(o),
}


def _pick(table, key, what):
    try:
        return table[key]
    except KeyError:
        raise ValueError('%s %r is not one of %s' % (what, key, sorted(table))) from None


def make_optimizer(cfg, params):
    return _pick(_OPTIMIZERS, cfg['schedule']['optimizer'], 'optimizer')(cfg['schedule'], params)


def make_loss(loss_type, cfg):
    return _pick(_LOSSES, loss_type, 'loss')()


def make_scheduler(optimizer, cfg):
    s = cfg['schedule']
    if not s['if_scheduler']:
        return None
    return _pick(_SCHEDULERS, s['scheduler'], 'scheduler')(optimizer, s, cfg)


def optim_hparams(cfg):
    """What the resident-scene engine needs to reproduce make_optimizer(cfg, ...): kind + the constructor arguments the
    reference passes (ADAM: lr; SGD: lr, momentum; RMSprop: lr, alpha — utils/utils.py:10-16), torch defaults otherwise."""
    s = cfg['schedule']
    kind = s['optimizer']
    _pick(_OPTIMIZERS, kind, 'optimizer')
    out = {'optimizer': kind, 'lr': float(s['lr']), 'betas': _ADAM_DEFAULTS[0], 'eps': _ADAM_DEFAULTS[1]}
    if kind == 'SGD':
        out['momentum'] = float(s['momentum'])
    if kind == 'RMSprop':
        out['alpha'] = float(s['alpha'])
    return out


def adam_hparams(cfg):
    """(lr, betas, eps) of the ADAM the reference constructs: lr from cfg, torch defaults otherwise."""
    if cfg['schedule']['optimizer'] != 'ADAM':
        raise ValueError('the fused HIP step implements ADAM only; got %s' % cfg['schedule']['optimizer'])
    return (float(cfg['schedule']['lr']),) + _ADAM_DEFAULTS


class _Schedule:
    """The per-epoch hyper-parameters the reference's own scheduler object would give its optimiser.

    The fused step takes lr / betas as launch arguments, so instead of re-deriving each of the eight scheduler kinds in
    closed form, the SAME torch scheduler (make_scheduler) is driven on a one-element dummy optimiser (make_optimizer: the
    same class and defaults) and its parameter group is read after every `scheduler.step()` — exactly the sequence the
    reference sees (mainsolver.py:60: one step per epoch).  OneCycleLR also cycles ADAM's beta1; that comes along."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.param = torch.nn.Parameter(torch.zeros(1))
        self.opt = make_optimizer(cfg, [self.param])
        self.sched = make_scheduler(self.opt, cfg)
        self.seq = [self._group()]

    def _group(self):
        g = self.opt.param_groups[0]
        return {k: (tuple(v) if isinstance(v, (tuple, list)) else v) for k, v in g.items() if k != 'params'}

    def at(self, epoch):
        while len(self.seq) <= epoch:
            if self.sched is not None:
                self.opt.step()                   # (keeps torch's "scheduler before optimizer" warning quiet; the gradient is None)
                self.sched.step()
            self.seq.append(self._group())
        return self.seq[epoch]


_SCHEDULES = {}


def epoch_hparams(cfg, epoch):
    """Optimiser parameter group (lr, betas, ...) in force during epoch `epoch` (0-based), any scheduler kind."""
    key = id(cfg)
    sch = _SCHEDULES.get(key)
    if sch is None or sch.cfg is not cfg:
        sch = _SCHEDULES[key] = _Schedule(cfg)
    return sch.at(epoch)


def epoch_lr(cfg, epoch):
    """Learning rate after `epoch` scheduler steps, for the fused step (every scheduler kind of make_scheduler)."""
    return float(epoch_hparams(cfg, epoch)['lr'])


def export_optimizer(cfg, params, flat_params, offsets, m, v, step_count, group=None):
    """The torch optimiser `make_optimizer(cfg, params)` would be after `step_count` steps, filled from the resident-scene
    engine's flat state vectors — so that `<time>_curweights.pth` holds what the reference's `save_checkpoint(model,
    optimizer)` stores (utils/utils.py:82-88) and `load_checkpoint` (:91-102) can feed it to `make_optimizer(cfg)` again.
    Per optimiser kind (torch's own state keys):
      ADAM     m -> exp_avg, v -> exp_avg_sq, step
      SGD      m -> momentum_buffer (only when momentum != 0 and a step has been taken, as torch creates it)
      RMSprop  m -> square_avg, step
    params: `model.parameters()` — the order the reference hands to make_optimizer, which is the order a state_dict numbers
    them in; flat_params / offsets: the same nn.Parameters in flat-vector order with their start in m / v; group:
    hyper-parameters in force (lr after the scheduler, OneCycleLR's betas / momentum) written into the parameter group."""
    opt = make_optimizer(cfg, list(params))
    kind = cfg['schedule']['optimizer']
    if group:
        for k, val in group.items():
            if k in opt.param_groups[0] and k != 'params':
                opt.param_groups[0][k] = val
    for i, p in enumerate(flat_params):
        n = p.numel()
        seg_m = m[offsets[i]:offsets[i] + n].view(p.shape).clone()
        if step_count <= 0:                       # torch creates an optimiser's state on its first step
            continue
        if kind == 'ADAM':
            opt.state[p] = {'step': torch.tensor(float(step_count)), 'exp_avg': seg_m,
                            'exp_avg_sq': v[offsets[i]:offsets[i] + n].view(p.shape).clone()}
        elif kind == 'SGD':
            if opt.param_groups[0]['momentum'] != 0:
                opt.state[p] = {'momentum_buffer': seg_m}
        elif kind == 'RMSprop':
            opt.state[p] = {'step': torch.tensor(float(step_count)), 'square_avg': seg_m}
    return opt


# ---------------------------------------------------------------------------------------------- checkpoints
def _bundle(model, optimizer, **extra):
    d = {'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}
    d.update(extra)
    return d


def save_checkpoint(model, optimizer, filename='my_checkpoint.pth.tar'):
    torch.save(_bundle(model, optimizer), filename)


def save_point_sche(model, optimizer, schedule, filename='my_checkpoint.pth.tar'):
    torch.save(_bundle(model, optimizer, schedule=schedule.state_dict()), filename)


def load_model(checkpoint_file, model, device):
    """Weights only (a file this code or the reference wrote; loaded without unpickling arbitrary objects)."""
    state = torch.load(checkpoint_file, map_location=device, weights_only=True)
    model.load_state_dict(state['state_dict'], strict=False)
    return state


def load_checkpoint(checkpoint_file, model, optimizer, lr, device):
    state = load_model(checkpoint_file, model, device)
    optimizer.load_state_dict(state['optimizer'])
    for group in optimizer.param_groups:
        group['lr'] = lr


def seed_everything(seed=42):
    os.environ['PYTHONHASHSEED'] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
