import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dual-modal-fusion_amd'), os.path.join(ROOT, 'tests'), ROOT]
from test_gpu_parity import SHAPES, make_cfg
from dmf.engine import QuaScene, QuaTrainEngine
from oracle.solver_ref import qua_train_steps
from oracle.gmfnet_ref import Net as RefNet
from model.gmfnet import Net as HipNet
C, C2, P, S, K = SHAPES['qua']
cfg = make_cfg('qua'); cfg['gmf']['single_input'] = 1
dqtl = {'alpha': 1.0, 'beta': 0.5, 'gamma': 0.5, 'epsilon': 1e-8, 'tao': 2.0}
g = torch.Generator().manual_seed(6)
H, W, bs, n = 20, 18, 8, 6
scenes = [(torch.rand(H + P - 1, W + P - 1, C, generator=g) - 0.2).numpy() for _ in range(4)]
xy = torch.stack([torch.randint(0, H, (n * bs,), generator=g), torch.randint(0, W, (n * bs,), generator=g)], 1).int()
lab = torch.randint(0, K, (n * bs,), generator=g)
def cpu_run(thr):
    torch.set_num_threads(thr)
    torch.manual_seed(5)
    ref = RefNet(cfg)
    opt = torch.optim.RMSprop(ref.parameters(), lr=2e-3, alpha=0.9)
    traj = []
    for i in range(n):
        qua_train_steps(ref, scenes, xy.numpy(), lab.numpy(), bs, P, dqtl, optimizer=opt, batches=[np.arange(i * bs, (i + 1) * bs)])
        traj.append({k: v.clone() for k, v in ref.state_dict().items()})
    return traj
cpus = {t: cpu_run(t) for t in (1, 4, 16)}
torch.manual_seed(5)
ref = RefNet(cfg)
hip = HipNet(cfg); hip.load_state_dict(ref.state_dict()); hip = hip.cuda()
eng = QuaTrainEngine(hip, QuaScene(scenes, 'cuda:0'), bs, dqtl, optimizer='RMSprop', lr=2e-3, alpha=0.9)
eng.load_plan(xy, lab)
gtraj = []
for i in range(n):
    eng.run_plan(1, 0)
    torch.cuda.synchronize()
    gtraj.append({k: v.detach().cpu().clone() for k, v in hip.state_dict().items()})
for i in range(n):
    line = 'step %d:' % (i + 1)
    for a, b, nm in ((cpus[1], cpus[4], 'cpu1-cpu4'), (cpus[1], cpus[16], 'cpu1-cpu16'), (cpus[1], gtraj, 'cpu1-gpu'), (cpus[16], gtraj, 'cpu16-gpu')):
        d = max(float((a[i][k] - b[i][k]).abs().max()) for k in a[i] if k != 'pool_w')
        line += '  %s %.2e' % (nm, d)
    print(line)
k = 'spec_a.weight'
d = (cpus[1][n - 1][k] - gtraj[n - 1][k]).abs().reshape(40, -1)
print('spec_a.weight |cpu1 - gpu| per channel max:', ['%.1e' % x for x in d.max(1).tolist()])
