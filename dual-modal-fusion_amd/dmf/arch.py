"""GMFNet hyper-parameters from a reference-style cfg dict.

Reference keys consumed: `patch_size`, `Categories_Number`, `DATA_DICT[data_city].size[2]`
(config.yml:27-28,77-80), `trans.embed_dim` / `trans.num_head` (config.yml:66-73; consumed by nothing in the
reference).  Keys this build adds: `scale` (aux/primary resolution ratio; the reference hard-codes 4,
train/dataset.py:166), `aux_bands`, `gmf.{width,groups,hidden,pool_sigma,attention,single_input,half}`.
"""
import math

import torch


def auto_groups(C, width):
    """Largest G <= 16 with C % G == 0, width % G == 0, whole 4-channel blocks per group ((width/G) % 4 == 0) and band groups
    the kernel reads as whole 16-byte chunks ((C/G) % 4 == 0)."""
    best = 1
    for g in range(1, 17):
        if C % g == 0 and width % g == 0 and (width // g) % 4 == 0 and (C // g) % 4 == 0:
            best = g
    return best


def arch_from_cfg(cfg):
    gmf = dict(cfg.get('gmf') or {})
    trans = dict(cfg.get('trans') or {})
    C = int(cfg['DATA_DICT'][cfg['data_city']]['size'][2])
    width = int(gmf.get('width', 40))
    groups = gmf.get('groups', 'auto')
    if groups in ('auto', None, 0):
        groups = auto_groups(C, width)
    single = int(gmf.get('single_input', 0))     # stage-2 net (tostagesolver.py:274): one input, aux = its band mean
    return dict(
        single_input=single,
        C=C, C2=1 if single else int(cfg.get('aux_bands', 1)), P=int(cfg['patch_size']),
        S=1 if single else int(cfg.get('scale', 4)),
        K=int(cfg['Categories_Number']), F=width, G=int(groups), H=int(gmf.get('hidden', 64)),
        sigma=float(gmf.get('pool_sigma', 2.5)), attention=int(gmf.get('attention', 0)),
        heads=int(trans.get('num_head', 3)), E=int(trans.get('embed_dim', 96)),
        half=int(gmf.get('half', 0)),                # fp16 primary scene / patches, fp16 operands in spec_a (include/dmf.h: dmf_input.half)
    )


def anchor_pool_weights(P, sigma):
    """Fixed pooling profile centred on patch element (0,0): the reference anchors a pixel's patch at its
    top-left corner (function.py:103-110, dataset.py:175).  sigma <= 0 -> uniform mean."""
    r = torch.arange(P, dtype=torch.float64)
    if sigma > 0:
        w = torch.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / (2.0 * sigma * sigma))
    else:
        w = torch.ones(P, P, dtype=torch.float64)
    return (w / w.sum()).reshape(-1).to(torch.float32)
