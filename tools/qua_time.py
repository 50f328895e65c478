import sys, os, torch, numpy as np
ROOT='/root/repo'
sys.path[:0]=[os.path.join(ROOT,'dual-modal-fusion_amd'), ROOT]
from dmf import lib
bs, K = 256, 12
g = torch.Generator().manual_seed(0)
logits = torch.randn(4*bs, K, generator=g).cuda()
lab = torch.randint(0, K, (bs,), generator=g).int().cuda()
loss = torch.zeros(1, device='cuda'); dl = torch.empty_like(logits)
def run(name, prm, dlogits):
    for _ in range(20): lib.qua_loss(logits, bs, lab, prm, loss=loss, dlogits=dlogits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): lib.qua_loss(logits, bs, lab, prm, loss=loss, dlogits=dlogits)
    e1.record(); torch.cuda.synchronize()
    print('%-40s %.2f us per call' % (name, e0.elapsed_time(e1) / 200 * 1e3))
full = lib.QuaParams(alpha=0.1, beta=0.05, gamma=1.0, epsilon=1e-8, tao=0.1)
nob = lib.QuaParams(alpha=0.1, beta=0.0, gamma=1.0, epsilon=1e-8, tao=0.1)
run('full', full, dl)
run('loss only (no sweep 3)', full, None)
run('beta = 0 (no sweep 2)', nob, dl)
run('beta = 0, loss only', nob, None)
