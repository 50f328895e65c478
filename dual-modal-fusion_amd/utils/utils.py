"""utils.utils — optimiser / loss / scheduler factories and checkpoint helpers (mirror of the reference module).

`make_optimizer`, `make_loss`, `make_scheduler` build the same torch objects from the same cfg keys as the
reference (utils/utils.py:8-71): ADAM = `torch.optim.Adam(params, lr)` with every other default, `Criterion` =
`nn.CrossEntropyLoss()`, the eight scheduler kinds.  They serve the drop-in path in which the reference-style
solver loop drives `model.gmfnet.Net` through autograd; the resident-scene fast path (dmf/engine.py) applies the
same Adam update in `dmf_grad_reduce_adam` and takes only the hyper-parameters from here (`adam_hparams`).
Checkpoint file formats are the reference's (:82-111): `{'state_dict', 'optimizer'}`.
"""
import os
import random

import numpy as np
import torch
import torch.nn as nn
import torch.optim.lr_scheduler as lr_scheduler


def make_optimizer(cfg, params):
    opt_type = cfg['schedule']['optimizer']
    if opt_type == "ADAM":
        return torch.optim.Adam(params, lr=cfg['schedule']['lr'])
    if opt_type == "SGD":
        return torch.optim.SGD(params, lr=cfg['schedule']['lr'], momentum=cfg['schedule']['momentum'])
    if opt_type == "RMSprop":
        return torch.optim.RMSprop(params, lr=cfg['schedule']['lr'], alpha=cfg['schedule']['alpha'])
    raise ValueError(opt_type)


def adam_hparams(cfg):
    """(lr, betas, eps) of the ADAM the reference constructs: lr from cfg, torch defaults otherwise."""
    if cfg['schedule']['optimizer'] != "ADAM":
        raise ValueError('the fused HIP step implements ADAM only; got %s' % cfg['schedule']['optimizer'])
    return float(cfg['schedule']['lr']), (0.9, 0.999), 1e-8


def make_loss(loss_type, cfg):
    if loss_type == "MSE":
        return nn.MSELoss(reduction='mean')
    if loss_type == "L1":
        return nn.L1Loss(reduction='mean')
    if loss_type == "Criterion":
        return nn.CrossEntropyLoss()
    if loss_type == "KL":
        return nn.KLDivLoss(reduction='batchmean')
    if loss_type == 'qua_loss':
        from train.loss_function import qua_loss
        return qua_loss()
    raise ValueError(loss_type)


def make_scheduler(optimizer, cfg):
    sch = cfg['schedule']
    if not sch['if_scheduler']:
        return None
    kind = sch['scheduler']
    if kind == "StepLR":
        return lr_scheduler.StepLR(optimizer, step_size=50, gamma=sch['base_lr'] / sch['lr'])
    if kind == "LinearLR":
        return lr_scheduler.LinearLR(optimizer, start_factor=0.1, end_factor=1, total_iters=10)
    if kind == "CosineAnnealingLR":
        return lr_scheduler.CosineAnnealingLR(optimizer, 50, sch['base_lr'])
    if kind == "CyclicLR":
        return lr_scheduler.CyclicLR(optimizer, base_lr=sch['base_lr'], max_lr=sch['lr'], step_size_up=10,
                                     step_size_down=40, cycle_momentum=False)
    if kind == "OneCycleLR":
        return lr_scheduler.OneCycleLR(optimizer, max_lr=sch['lr'], pct_start=0.5, total_steps=cfg['epoch'],
                                       div_factor=sch['lr'] / sch['base_lr'], final_div_factor=sch['lr'] / sch['base_lr'])
    if kind == "ConstantLR":
        return lr_scheduler.ConstantLR(optimizer, factor=sch['base_lr'] / sch['lr'], total_iters=10)
    if kind == "ChainedScheduler":
        return lr_scheduler.ChainedScheduler([lr_scheduler.LinearLR(optimizer, start_factor=0.1, end_factor=1, total_iters=10),
                                              lr_scheduler.ExponentialLR(optimizer, gamma=0.98)])
    if kind == "ExponentialLR":
        return lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=0.98)
    raise ValueError(kind)


def epoch_lr(cfg, epoch):
    """Learning rate the reference's scheduler yields after `epoch` scheduler steps, for the fused step
    (ExponentialLR only; other kinds go through the torch optimiser of the drop-in path)."""
    sch = cfg['schedule']
    if not sch['if_scheduler']:
        return float(sch['lr'])
    if sch['scheduler'] != 'ExponentialLR':
        raise ValueError('the fused HIP step supports ExponentialLR only; got %s' % sch['scheduler'])
    return float(sch['lr']) * 0.98 ** epoch


def save_point_sche(model, optimizer, schedule, filename="my_checkpoint.pth.tar"):
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                "schedule": schedule.state_dict()}, filename)


def save_checkpoint(model, optimizer, filename="my_checkpoint.pth.tar"):
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}, filename)


def load_checkpoint(checkpoint_file, model, optimizer, lr, device):
    checkpoint = torch.load(checkpoint_file, map_location=device, weights_only=True)
    model.load_state_dict(checkpoint["state_dict"], strict=False)
    optimizer.load_state_dict(checkpoint["optimizer"])
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr


def load_model(checkpoint_file, model, device):
    checkpoint = torch.load(checkpoint_file, map_location=device, weights_only=True)
    model.load_state_dict(checkpoint["state_dict"], strict=False)


def seed_everything(seed=42):
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
