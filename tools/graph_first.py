import sys, os, time, ctypes, torch, numpy as np
ROOT='/root/repo'
sys.path[:0]=[os.path.join(ROOT,'dual-modal-fusion_amd'), ROOT]
import bench
class A: pass
args = A(); args.config='1'
for k,v in bench.CONFIGS['1'].items():
    if k!='name': setattr(args,k,v)
args.aux_bands=1; args.half=0; args.train_rate=0.1
from dmf.engine import Scene, TrainEngine
from model.gmfnet import Net
cfg = bench.make_cfg(args)
MS, PAN, xy_tab, lab_tab, train, test = bench.build_problem(args, cfg)
torch.manual_seed(0)
net = Net(cfg).cuda()
eng = TrainEngine(net, Scene(MS, PAN, 'cuda:0'), 256)
idx = bench.make_plan(train, 400, 256, seed=1)
eng.load_plan(xy_tab[idx], lab_tab[idx])
bench.prewarm(eng, net, args, 60.0)
eng.run_plan(5, 0)
eng._capture(20)
hip = ctypes.CDLL('libamdhip64.so')
up = len(sys.argv) > 1
if up:
    try:
        ex = eng.graph.raw_cuda_graph_exec()
        rc = hip.hipGraphUpload(ctypes.c_void_p(ex), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        print('upload rc', rc)
    except Exception as e:
        print('upload failed', e)
torch.cuda.synchronize()
for r in range(4):
    t0 = time.perf_counter(); eng.run_plan(20, 20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('replay %d: %.1f us/step' % (r, dt / 20 * 1e6))
