"""Launcher (mirror of the reference test.py:7-14): seed 3407, render config.yml, Solver(cfg).run().

Under `python -m torch.distributed.run --nproc-per-node N test.py config.yml` (one process per GPU) the same run is
data parallel: every rank seeds identically, takes GPU LOCAL_RANK, and the solver shards every batch (solver/mainsolver.py).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from solver.mainsolver import Solver          # noqa: E402
from utils.config import get_render_config    # noqa: E402


def attach_process_group(solver, cfg):
    """Give the solver the job's process group and, when it can be proven, the one-shot gradient exchange."""
    import torch.distributed as dist
    from dmf import xgmi
    solver.process_group = dist.group.WORLD
    solver.rank, solver.world = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() == 'nccl' and cfg.get('xgmi_exchange', 1):
        import importlib
        net = importlib.import_module('model.' + cfg['model_name'].lower()).Net(args=cfg)
        solver.comm = xgmi.create(sum(p.numel() for p in net.parameters()), solver.process_group)
    return solver


if __name__ == "__main__":
    torch.manual_seed(3407)
    world = int(os.environ.get('WORLD_SIZE', 1))
    path = sys.argv[1] if len(sys.argv) > 1 else "config.yml"
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('DMF_DIST_BACKEND', 'nccl'))
        # rank 0 renders the configuration (it numbers and creates the run directory); the others receive its result
        box = [get_render_config(path) if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        cfg = box[0]
        cfg['device'] = 'cuda:%d' % local
        torch.manual_seed(3407)
        attach_process_group(Solver(cfg), cfg).run()
        dist.destroy_process_group()
    else:
        Solver(get_render_config(path)).run()
