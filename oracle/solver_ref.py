"""ORACLE — test infrastructure only (tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke()).

CPU restatement of the reference's training / evaluation loop bodies around the oracle Net, with the
aux/primary resolution ratio generalised from the hard-coded 4 to `S`:
  * batch materialisation  — train/dataset.py:168-185 (slice HWC window, transpose to CHW, float32)
  * train step             — solver/mainsolver.py:49-55 (zero_grad, forward, CE(target.long()), backward, Adam.step)
  * eval                   — solver/mainsolver.py:102-141 (argmax, test_matrix[pred][target] += 1)
  * stage-2 step / eval     — solver/tostagesolver.py:268-278 (concat four streams, one-input net, qua_loss) and
                              :331-341 (argmax of softmax(out[:bs] + out[bs:2bs]))
Its equivalence with the REAL reference solvers is pinned by tests/golden/g9_trajectory.npz (Solver, S = 4) and
tests/golden/g10_stage2.npz (toStageSolver) in tests/test_golden_trajectory.py.
"""
import numpy as np
import torch

from oracle import datapath_ref as dref


def materialise(MS, PAN, xy, P, S):
    a = np.stack([MS[x:x + P, y:y + P, :].transpose(2, 0, 1) for x, y in xy])
    if PAN.ndim == 2:
        b = np.stack([PAN[S * x:S * x + S * P, S * y:S * y + S * P][None] for x, y in xy])
    else:
        b = np.stack([PAN[S * x:S * x + S * P, S * y:S * y + S * P, :].transpose(2, 0, 1) for x, y in xy])
    return torch.from_numpy(np.ascontiguousarray(a)).type(torch.FloatTensor), \
        torch.from_numpy(np.ascontiguousarray(b)).type(torch.FloatTensor)


def train_steps(net, MS, PAN, xy_plan, label_plan, B, P, S, lr=1e-3, optimizer=None, on_step=None):
    """Runs len(xy_plan)//B reference train steps; returns (per-step losses, optimizer)."""
    opt = optimizer or torch.optim.Adam(net.parameters(), lr=lr)     # utils/utils.py:12
    ce = torch.nn.CrossEntropyLoss()                                  # utils/utils.py:29
    net.train()
    losses = []
    for i in range(0, len(xy_plan) - B + 1, B):
        a, b = materialise(MS, PAN, xy_plan[i:i + B], P, S)
        target = torch.from_numpy(np.asarray(label_plan[i:i + B], dtype=np.float32))
        opt.zero_grad()
        out = net(a, b)
        loss = ce(out, target.long())
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if on_step is not None:
            on_step(len(losses))
    return losses, opt


def evaluate(net, MS, PAN, xy, labels, K, P, S, batch=300):
    net.eval()
    m = np.zeros([K, K])
    logits_all = []
    with torch.no_grad():
        for i in range(0, len(xy), batch):
            a, b = materialise(MS, PAN, xy[i:i + batch], P, S)
            out = net(a, b)
            logits_all.append(out)
            pred = out.data.max(1, keepdim=True)[1]
            m += dref.confusion(pred.numpy(), labels[i:i + batch], K)
    return m, torch.cat(logits_all)


# ---------------------------------------------------------------------------------------------- stage 2
def materialise4(scenes, xy, P):
    """train/dataset.py:203-221 for a batch, concatenated like tostagesolver.py:272: [4*bs, C, P, P]."""
    parts = [np.stack([im[x:x + P, y:y + P, :].transpose(2, 0, 1) for x, y in xy]) for im in scenes]
    return torch.from_numpy(np.ascontiguousarray(np.concatenate(parts))).type(torch.FloatTensor)


def qua_train_steps(net, scenes, xy_plan, label_plan, bs, P, dqtl, lr=1e-3, optimizer=None, batches=None):
    """tostagesolver.py:268-278.  `batches` (optional) = explicit list of index arrays into the plan, else
    consecutive batches of `bs`."""
    opt = optimizer or torch.optim.Adam(net.parameters(), lr=lr)
    net.train()
    losses = []
    if batches is None:
        batches = [np.arange(i, i + bs) for i in range(0, len(xy_plan) - bs + 1, bs)]
    for idx in batches:
        data = materialise4(scenes, xy_plan[idx], P)
        target = torch.from_numpy(np.asarray(label_plan[idx], dtype=np.float32))
        n = len(idx)
        opt.zero_grad()
        out = net(data)
        loss = dref.qua_loss(out, n, target, dqtl['alpha'], dqtl['beta'], dqtl['gamma'], dqtl['epsilon'], dqtl['tao'])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses, opt


def qua_evaluate(net, scenes, xy, labels, K, P, batch=300):
    """tostagesolver.py:331-341 over all given pixels."""
    net.eval()
    m = np.zeros([K, K])
    with torch.no_grad():
        for i in range(0, len(xy), batch):
            n = len(xy[i:i + batch])
            out = net(materialise4(scenes, xy[i:i + batch], P))
            pred = (out[:n] + out[n:2 * n]).softmax(dim=-1).data.max(1, keepdim=True)[1]
            m += dref.confusion(pred.numpy(), labels[i:i + batch], K)
    return m
