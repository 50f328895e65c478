"""solver.tostagesolver — `toStageSolver`: the reference's two-stage path (solver/tostagesolver.py:20-414).

Status in this build: the pieces of that path which exist in the reference are built —
`pan2ms` (image_convert/IHS.py + the `dmf_pan2ms` HIP kernel), `qua_loss` (train/loss_function.py),
`dataset_qua_dqtl` (train/dataset.py) — but the path as a whole is not runnable yet: stage 1 trains
`model.generator` / `model.discriminator`, which the reference does not ship, and the stage-2 network
takes one four-stream input whose architecture is likewise absent.  The class exists so that
`from solver.tostagesolver import toStageSolver` resolves and fails with a clear message instead of an ImportError.
"""
from solver.mainsolver import Solver


class toStageSolver(Solver):
    def __init__(self, cfg):
        super().__init__(cfg)

    def train_stage1(self):
        raise NotImplementedError('stage 1 (GAN pre-fusion) needs model.generator / model.discriminator, which the '
                                  'reference does not ship; see DESIGN.md §7')

    def run(self):
        raise NotImplementedError('the two-stage solver is not built yet (DESIGN.md §7: next); available pieces: '
                                  'image_convert.IHS.pan2ms(_gpu), train.loss_function.qua_loss, train.dataset.dataset_qua_dqtl')
