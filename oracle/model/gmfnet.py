"""ORACLE — test infrastructure only.

Plugin shim that lets the REAL reference solver (`/root/reference/solver/mainsolver.py:31-34`,
`importlib.import_module('model.gmfnet').Net(args=cfg)`) drive the CPU oracle net.  Only
oracle/make_goldens.py puts this directory on sys.path (as `model`); the product's own plugin is
dual-modal-fusion_amd/model/gmfnet.py.
"""
import torch

from oracle.gmfnet_ref import Net as _RefNet

TRACE = {'enabled': False, 'init_state': None, 'logits': [], 'train_flags': []}


class Net(_RefNet):
    def __init__(self, args):
        super().__init__(args)
        if TRACE['enabled'] and TRACE['init_state'] is None:
            TRACE['init_state'] = {k: v.detach().clone() for k, v in self.state_dict().items()}

    def forward(self, a, b=None):
        out = super().forward(a, b)
        if TRACE['enabled']:
            TRACE['logits'].append(out.detach().clone())
            TRACE['train_flags'].append(bool(self.training))
        return out
