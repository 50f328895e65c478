"""A/B of the two forms of the gradient-reduce launch (DMF_REDUCE_V1=1 selects the first): every gradient bit must agree
(same chunking, same summation order), then the warm back-to-back time of each.  GPU box only.

    python tools/reduce_ab.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), os.path.join(ROOT, 'tests'), ROOT]


def main():
    from dmf import lib
    from test_gpu_parity import SHAPES, nets, rand_batch
    bad = 0
    for name, B in (('hsi', 256), ('hsi', 64), ('hsi', 300), ('hsi', 700), ('tiny', 33), ('tiny', 7), ('panms', 260), ('qua', 1100),
                    ('hsi224', 256), ('hsi_attn', 64)):
        if name not in SHAPES:
            continue
        cfg, ref, hip = nets(name)
        a, b, t = rand_batch(name, B)
        K = SHAPES[name][4]
        ad, bd = a.cuda(), b.cuda()
        inp = lib.input_patches(hip.shape, ad, bd)
        theta = hip.flat_parameters()
        logits = torch.empty(B, K, device='cuda'); loss = torch.empty(B, device='cuda')
        ws = hip.workspace(B)
        if hip.shape.attention:
            aws = torch.empty(lib.attn_train_workspace_bytes(hip.shape, B), dtype=torch.uint8, device='cuda')
            lib.train_attn_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), None, 1.0 / B, logits, loss, ws, aws)
        else:
            lib.train_fwd_bwd(hip.shape, inp, theta, hip.pool_w, t.int().cuda(), 1.0 / B, logits, loss, ws)
        out = {}
        for v1 in ('1', '0'):
            os.environ['DMF_REDUCE_V1'] = v1
            g = torch.full_like(theta, float('nan'))
            lib.grad_reduce(hip.shape, B, ws, g)
            th = theta.clone(); m = torch.zeros_like(th); v = torch.zeros_like(th)
            step = torch.full((1,), 1234, dtype=torch.int32, device='cuda')
            lib.grad_reduce_adam(hip.shape, B, ws, th, m, v, None, 1e-3, 0.9, 0.999, 1e-8, 0, adam_step_dev=step)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(200):
                lib.grad_reduce_adam(hip.shape, B, ws, th, m, v, None, 1e-3, 0.9, 0.999, 1e-8, 0, adam_step_dev=step)
            ev[1].record()
            torch.cuda.synchronize()
            out[v1] = (g.clone(), th.clone(), ev[0].elapsed_time(ev[1]) / 200 * 1e3)
        same_g = torch.equal(out['1'][0], out['0'][0])
        d_th = (out['1'][1] - out['0'][1]).abs().max().item()
        n_diff = int((out['1'][0] != out['0'][0]).sum())
        print('%-9s B=%-5d n=%-6d gradient bits equal: %s (%d differ)  max|theta_v1 - theta_v2| after 201 ADAM steps %.2e   '
              'warm back-to-back: v1 %.2f us  v2 %.2f us' % (name, B, theta.numel(), same_g, n_diff, d_th, out['1'][2], out['0'][2]), flush=True)
        bad += 0 if same_g else 1
    os.environ.pop('DMF_REDUCE_V1', None)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
