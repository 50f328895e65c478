"""configs[1] at full size: "class maps and kappa identical to the reference CPU path on the same patches" (BASELINE.json).

The 145 x 145 x 200 synthetic scene of bench.py, 2,200 fused train steps of batch 256 on the GPU, then EVERY one of the
21,025 pixels is classified twice with the SAME weights: by the HIP eval path (dmf_forward through EvalEngine, on-device
confusion matrix and label map) and by the CPU oracle's forward.  Logits within 1e-5; the class map identical wherever the
oracle's top-2 margin exceeds 1e-4 (an argmax cannot be asked to survive a tie inside the logit tolerance); the confusion
matrix and kappa over the labelled pixels identical.  (Two independent fp32 TRAININGS drift apart chaotically — bench.py
reports that with its noise floor; the parity statement is this one: same weights, same patches.)"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_full_scene_class_map_and_confusion_match_the_oracle_on_the_same_weights():
    from dmf import synth
    from dmf.engine import EvalEngine, Scene, TrainEngine
    from function.function import data_padding, data_padding_aux, split_data_old
    from indicators.kappa import aa_oa_quiet
    from model.gmfnet import Net
    from oracle.gmfnet_ref import Net as RefNet
    from oracle import solver_ref
    import contextlib, io
    H = W = 145
    C, P, K, B, STEPS = 200, 11, 17, 256, 2200
    cfg = {'patch_size': P, 'Categories_Number': K, 'data_city': 'syn', 'DATA_DICT': {'syn': {'size': [H, W, C], 'color': synth.class_colors(K)}},
           'scale': 1, 'aux_bands': 1, 'gmf': {'width': 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0}}
    primary, aux, label = synth.make_scene(H, W, C, 1, 1, n_classes=K - 1, seed=0)
    MS = data_padding(primary, cfg, 'ms').astype(np.float32)
    PAN = data_padding_aux(aux, cfg).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        xyl, idx = split_data_old(label, cfg)
    xy_tab = np.concatenate([xyl[0], xyl[1]], 1).astype(np.int32)            # all 21,025 pixels, row-major (function.py:149-169)
    lab_tab = xyl[2].reshape(-1).astype(np.int32)
    labelled = np.array(idx[1])
    g = torch.Generator().manual_seed(3407)
    perm = torch.randperm(len(labelled), generator=g).numpy()
    train = labelled[perm[:int(0.10 * len(labelled))]]
    plan = np.concatenate([train[torch.randperm(len(train), generator=g).numpy()] for _ in range(STEPS * B // len(train) + 1)])[:STEPS * B]
    torch.manual_seed(3407)
    net = Net(cfg).cuda()
    scene = Scene(MS, PAN, 'cuda:0')
    eng = TrainEngine(net, scene, B, lr=1e-3)
    eng.load_plan(xy_tab[plan], lab_tab[plan])
    eng.run_plan(STEPS, 50)
    losses = eng.mean_losses().numpy()
    assert np.isfinite(losses).all() and losses[-50:].mean() < 0.5 * losses[:50].mean()      # it did train

    ev = EvalEngine(net, scene, 2048)
    m_gpu = ev.confusion(xy_tab[labelled], lab_tab[labelled]).cpu().numpy().astype(np.float64)
    map_gpu = ev.label_map(xy_tab, H, W).cpu().numpy()
    logits_gpu = torch.cat([ev.predict(torch.from_numpy(xy_tab[i:i + 2048]).cuda())[0].cpu().clone() for i in range(0, len(xy_tab), 2048)])

    ref = RefNet(cfg)
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    _, logits_cpu = solver_ref.evaluate(ref, MS, PAN, xy_tab, lab_tab, K, P, 1, batch=512)
    err = (logits_gpu - logits_cpu).abs().max().item()
    print('21,025 pixels, same weights after %d steps: max |logit difference| %.2e' % (STEPS, err))
    assert err < 1e-5
    pred_cpu = logits_cpu.argmax(1).numpy()
    map_cpu = np.zeros((H, W), dtype=np.int64)
    map_cpu[xy_tab[:, 0], xy_tab[:, 1]] = pred_cpu
    top2 = logits_cpu.topk(2, dim=1).values
    safe = ((top2[:, 0] - top2[:, 1]) > 1e-4).numpy().reshape(H, W)            # (xy_tab is row-major)
    assert (map_gpu[safe] == map_cpu[safe]).all()
    n_diff = int((map_gpu != map_cpu).sum())
    print('class maps: %d of %d pixels differ (%d pixels inside the 1e-4 top-2 margin)' % (n_diff, H * W, int((~safe).sum())))
    m_cpu = np.zeros((K, K))
    np.add.at(m_cpu, (pred_cpu[labelled], lab_tab[labelled]), 1)
    flips = int(np.abs(m_gpu - m_cpu).sum() // 2)
    k_gpu, k_cpu = aa_oa_quiet(m_gpu)[2], aa_oa_quiet(m_cpu)[2]
    print('confusion over %d labelled pixels: %d differing entries, kappa GPU %.6f CPU %.6f' % (len(labelled), flips, k_gpu, k_cpu))
    assert flips <= int((~safe).sum()) and abs(k_gpu - k_cpu) <= 1e-3
    if n_diff == 0:
        assert flips == 0 and k_gpu == k_cpu
