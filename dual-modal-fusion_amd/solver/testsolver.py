"""solver.testsolver — `Testsolver(cfg)`: the reference file is a truncated stub (solver/testsolver.py:9-15) whose
constructor loads `model.<cfg['algorithm']>` and takes `lib.Net`; this class completes exactly that and adds the
evaluation entry points one would expect from it (`test()` / `color()` inherited from Solver)."""
import importlib

from solver.mainsolver import Solver


class Testsolver(Solver):
    def __init__(self, cfg):
        cfg = dict(cfg)
        cfg.setdefault('model_name', cfg.get('algorithm', 'gmfnet'))
        cfg['model_name'] = cfg.get('algorithm', cfg['model_name'])
        super().__init__(cfg)
        lib = importlib.import_module("model." + self.cfg['model_name'].lower())
        self.net_class = lib.Net

    def run(self):
        while self.time < self.TIME:
            self.dataloader()
            self.test()
            if self.cfg['color']['index']:
                self.color()
            self.time += 1
