#!/bin/bash
# where an epoch of the solver's fast path goes, host side included: cProfile of Solver(cfg).train() on the synthetic
# 145x145x200 scene of config.yml (tools/solver_epoch_profile.sh [epochs])
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
R=/tmp/dmf_prof && rm -rf $R && mkdir -p $R/run/data/syn145
python3 tools/make_synthetic_scene.py $R/run/data/syn145/ > /dev/null 2>&1 || exit 1
cp -r dual-modal-fusion_amd $R/run/pkg && cd $R/run/pkg && python3 - "${1:-40}" <<'PY'
import cProfile, pstats, sys, time, torch
sys.path[:0] = ['.']
from utils.config import get_render_config
from solver.mainsolver import Solver
cfg = get_render_config('config.yml')
cfg['epoch'] = int(sys.argv[1])
torch.manual_seed(3407)
import torch._dynamo  # (its one-off import, which torch.optim triggers, is not an epoch's cost)
s = Solver(cfg); s.dataloader()
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
s.train()
torch.cuda.synchronize()
pr.disable(); t1 = time.time()
n_train = len(s.train_index_loader.dataset)
print('train: %d epochs of %d pixels in %.2f s -> %.2f ms / epoch, %.3f M patches/s end to end' % (
    cfg['epoch'], n_train, t1 - t0, (t1 - t0) / cfg['epoch'] * 1e3, cfg['epoch'] * n_train / (t1 - t0) / 1e6))
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats('mainsolver|basesolver|engine|utils/|dmf/lib|dataset|serialization|dataloader', 40)
st.sort_stats('tottime').print_stats(18)
PY
