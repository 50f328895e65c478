"""Launcher (mirror of the reference test.py:7-14): seed 3407, render config.yml, Solver(cfg).run()."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from solver.mainsolver import Solver          # noqa: E402
from utils.config import get_render_config    # noqa: E402

if __name__ == "__main__":
    torch.manual_seed(3407)
    cfg = get_render_config(sys.argv[1] if len(sys.argv) > 1 else "config.yml")
    Solver(cfg).run()
