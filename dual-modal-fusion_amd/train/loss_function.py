"""train.loss_function — `qua_loss`, the four-stream loss of the two-stage path (mirror of the reference module).

Same call signature and value as the reference (train/loss_function.py:15-76): input = logits of the four streams
stacked on the batch axis [4*bs, K], `t` = [bs] float class ids; six batch-mean KL terms with margin `tao`, the
balance term, and the class term against the SOFTMAX of the one-hot (:46-54).  Runs as torch ops on whatever
device the logits live on (a fused HIP version is listed as "next" in DESIGN.md); the one-hot is built with a
scatter instead of the reference's per-sample Python loop.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class qua_loss(nn.Module):
    def forward(self, out, bs, t, cfg):
        d = cfg['dqtl']
        alpha, beta, gamma, eps, tao = d['alpha'], d['beta'], d['gamma'], d['epsilon'], d['tao']
        data = out.softmax(dim=-1)
        p, q, r, s = data[:bs], data[bs:2 * bs], data[2 * bs:3 * bs], data[3 * bs:]

        def kl(log_in, tgt):
            return F.kl_div(log_in, tgt, reduction='batchmean')

        l1 = l2 = 0
        if alpha != 0:
            KL_M_GM, KL_M_GP = kl((r + eps).log(), p), kl((s + eps).log(), p)
            KL_P_GP, KL_P_GM = kl((r + eps).log(), q), kl((s + eps).log(), q)
            l1 = kl((q + eps).log(), p) + KL_M_GM + torch.abs(KL_M_GP - KL_M_GM + tao)
            l2 = kl((p + eps).log(), q) + KL_P_GP + torch.abs(KL_P_GM - KL_P_GP + tao)
        l3 = 0
        if beta != 0:
            KL_M_GP, KL_P_GM = kl((s + eps).log(), p), kl((s + eps).log(), q)
            l3 = torch.mean(torch.exp(-torch.abs(KL_M_GP / p)) + torch.exp(-torch.abs(KL_P_GM / q)))
        label = torch.zeros_like(p).scatter_(1, t.long().view(-1, 1), 1.0)
        l4 = kl((p + q).softmax(dim=-1).log(), label.softmax(dim=-1))
        return alpha * (l1 + l2) + beta * l3 + gamma * l4
