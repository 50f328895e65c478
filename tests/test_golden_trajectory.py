"""CPU: the oracle's restated loops (oracle/solver_ref.py — what bench.py times as `cpu_baseline` and what the GPU
parity tests compare against) reproduce the trajectories the REAL reference solvers produced around the oracle net:
  G9  Solver.train / Solver.test            (mainsolver.py:40-148)       tests/golden/g9_trajectory.npz
  G10 toStageSolver.train / .test, stage 2  (tostagesolver.py:259-346)   tests/golden/g10_stage2.npz
Both fixtures come from oracle/make_goldens.py, which imports the reference in the build container.
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import datapath_ref as dref      # noqa: E402
from oracle import solver_ref                # noqa: E402
from oracle.gmfnet_ref import Net            # noqa: E402


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    cfg = json.loads(str(g['cfg']))
    net = Net(cfg)
    net.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('init.')})
    return g, cfg, net


def _table(label):
    H, W = label.shape
    xs, ys = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    return np.stack([xs.reshape(-1), ys.reshape(-1)], 1), label.reshape(-1)


def _train_batches(g, B):
    """Cut the recorded index stream into the reference DataLoader's train batches.  Per epoch the stream holds
    n_train train visits followed by the validation visits the early-stopping loop made (mainsolver.py:62-76,
    tostagesolver.py:285-300); the two index sets are disjoint, which is enough to cut it."""
    n_train = len(g['split_train'])
    order = g['visit_order']
    valid = set(g['labelled'][g['split_valid']].tolist())
    pos, batches = 0, []
    while pos < len(order):
        ep = order[pos:pos + n_train]
        assert len(ep) == n_train and not (set(ep.tolist()) & valid)
        batches += [ep[i:i + B] for i in range(0, n_train, B)]
        pos += n_train
        while pos < len(order) and int(order[pos]) in valid:
            pos += 1
    return batches


def test_restated_train_loop_reproduces_reference_solver(golden_dir):
    g, cfg, net = _load(golden_dir, 'g9_trajectory.npz')
    P, S, B = cfg['patch_size'], cfg['scale'], cfg['batchsize']
    MS, PAN = dref.data_padding(g['primary'], P, S), dref.data_padding(g['aux'], P, S)
    xy, lab = _table(g['label'])
    n_train = len(g['split_train'])
    want = g['losses'][g['is_train_call']]
    batches = _train_batches(g, B)
    assert len(batches) == len(want)
    opt = torch.optim.Adam(net.parameters(), lr=cfg['schedule']['lr'])
    got = []
    for idx in batches:
        l, opt = solver_ref.train_steps(net, MS, PAN, xy[idx], lab[idx], len(idx), P, S, optimizer=opt)
        got += l
    assert np.abs(np.array(got) - want).max() < 1e-6
    for k, v in net.state_dict().items():
        if k != 'pool_w':
            assert np.allclose(v.numpy(), g['last.' + k], atol=1e-6), k


def test_restated_eval_reproduces_reference_test_matrix(golden_dir):
    g, cfg, _ = _load(golden_dir, 'g9_trajectory.npz')
    net = Net(cfg)
    net.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('best.')})
    P, S = cfg['patch_size'], cfg['scale']
    MS, PAN = dref.data_padding(g['primary'], P, S), dref.data_padding(g['aux'], P, S)
    xy, lab = _table(g['label'])
    first = g['labelled'][g['split_test']][:cfg['test_batchsize']]     # the reference stops after one batch (:142)
    m, logits = solver_ref.evaluate(net, MS, PAN, xy[first], lab[first], cfg['Categories_Number'], P, S)
    assert np.array_equal(m, g['test_matrix'])
    assert np.abs(logits.numpy() - g['test_logits']).max() < 1e-6
    assert abs(dref.aa_oa(m)[2] - float(g['kappa'])) < 1e-12


def test_restated_stage2_loop_reproduces_reference_tostagesolver(golden_dir):
    g, cfg, net = _load(golden_dir, 'g10_stage2.npz')
    P, B = cfg['patch_size'], cfg['batchsize']
    scenes = [dref.data_padding(g[k], P) for k in ('primary', 'pan4', 'ms_gan', 'pan_gan')]
    xy, lab = _table(g['label'])
    want = g['losses'][g['is_train_call']]
    assert net.arch['single_input'] == 1 and net.arch['S'] == 1 and net.arch['C2'] == 1
    batches = _train_batches(g, B)
    assert len(batches) == len(want)
    got, _ = solver_ref.qua_train_steps(net, scenes, xy, lab, B, P, cfg['dqtl'], lr=cfg['schedule']['lr'], batches=batches)
    assert np.abs(np.array(got) - want).max() < 1e-6
    for k, v in net.state_dict().items():
        if k != 'pool_w':
            assert np.allclose(v.numpy(), g['last.' + k], atol=1e-6), k
    # eval with the best weights: all test batches (tostagesolver.py:331-341 has no early stop)
    best = Net(cfg)
    best.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('best.')})
    test = g['labelled'][g['split_test']]
    m = solver_ref.qua_evaluate(best, scenes, xy[test], lab[test], cfg['Categories_Number'], P, batch=cfg['test_batchsize'])
    assert np.array_equal(m, g['test_matrix'])
    assert abs(dref.aa_oa(m)[2] - float(g['kappa'])) < 1e-12
