"""train.loss_function — `qua_loss`, the four-stream loss of the two-stage path (mirror of the reference module).

Same call signature and value as the reference (train/loss_function.py:15-76): input = logits of the four streams
stacked on the batch axis [4*bs, K], `t` = [bs] float class ids; six batch-mean KL terms with margin `tao`, the
balance term, and the class term against the SOFTMAX of the one-hot (:46-54).  Value and gradient come from one
launch of the `dmf_qua_loss` HIP kernel (include/dmf.h) behind a torch.autograd.Function; like the network it runs
on the GPU only.
"""
import torch
import torch.nn as nn

from dmf import lib


class _QuaFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, bs, labels, params):
        logits = out.contiguous().float()
        loss = torch.empty(1, device=out.device)
        dlogits = torch.empty_like(logits)
        lib.qua_loss(logits, bs, labels, params, loss=loss, dlogits=dlogits)
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None, None


class qua_loss(nn.Module):
    def forward(self, out, bs, t, cfg):
        if not out.is_cuda:
            raise lib.DmfError('train.loss_function.qua_loss runs on the GPU only (dmf_qua_loss HIP kernel)')
        labels = t.to(device=out.device).to(torch.int32).contiguous()
        return _QuaFunction.apply(out, int(bs), labels, lib.qua_params(cfg['dqtl']))
