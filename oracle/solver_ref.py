"""ORACLE — test infrastructure only (tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke()).

CPU restatement of the reference's training / evaluation loop bodies around the oracle Net, with the
aux/primary resolution ratio generalised from the hard-coded 4 to `S`:
  * batch materialisation  — train/dataset.py:168-185 (slice HWC window, transpose to CHW, float32)
  * train step             — solver/mainsolver.py:49-55 (zero_grad, forward, CE(target.long()), backward, Adam.step)
  * eval                   — solver/mainsolver.py:102-141 (argmax, test_matrix[pred][target] += 1)
Its equivalence with the REAL reference solver is pinned at S = 4 by tests/golden/g9_trajectory.npz
(tests/test_golden_trajectory.py).
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
