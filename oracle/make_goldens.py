"""ORACLE tooling — generates tests/golden/* by importing the REAL reference (build container only).

Run:  python oracle/make_goldens.py          (needs /root/reference; never runs on the GPU box)

What is imported from /root/reference and how
  * import cleanly as shipped: image_convert.IHS, train.loss_function, utils.utils
  * import after registering EMPTY placeholder modules for third-party packages this image lacks
    (cv2, libtiff, h5py, openpyxl, prefetch_generator): function.function, train.dataset,
    indicators.kappa, solver.basesolver, solver.mainsolver.  The placeholders carry NO functionality:
    any reference function that would call into them (read_tif, data_padding's cv2.copyMakeBorder,
    expo_result's Workbook) is NOT exercised and is NOT pinned by these fixtures.
  * never imported: the network (absent from the reference), train/train.py, train/test.py,
    train/verify.py (dead, not importable), solver/tostagesolver.py (needs absent model.generator).

Fixtures (SURVEY.md §8c list):
  g1_to_tensor.npz        to_tensor on a 6x7x5 cube                         (function.py:120-124)
  g2_split.npz            split_data_old tables on a 7x9 label map           (function.py:149-169)
  g2b_split_new.npz       split_data (data_new: 1) on fixed train / test masks  (function.py:172-194)
  g3_dataset.npz          dataset_dual[i] tuples                             (dataset.py:158-188)
  g5_ce_adam.npz          CrossEntropyLoss value/grad, Adam one step, ExponentialLR (utils/utils.py)
  g6_kappa.json           kappa / aa_oa on fixed matrices                    (kappa.py:10-22,69-84)
  g7_ihs.npz              unsampling / pan2ms / IHS_tran                     (IHS.py:6-54)
  g8_qua_loss.npz         qua_loss value + grad                              (loss_function.py:15-76)
  g9_trajectory.npz       reference Solver.train/test driving the oracle Net (mainsolver.py:40-148),
                          incl. the random_split / shuffle index stream under torch.manual_seed(3407) (G4)
  g10_stage2.npz          reference toStageSolver.train/test (tostagesolver.py:259-346) driving the oracle Net in
                          its single-input form over the real dataset_qua_dqtl and the real qua_loss.
                          solver/tostagesolver.py imports model.generator / model.discriminator (absent from the
                          reference) and torchvision (absent from the image) at module level: inert placeholders are
                          registered for them too; stage 1 (the GAN, tostagesolver.py:86-238) is NOT exercised —
                          its outputs ms_gan / pan_gan are synthetic arrays, as with dqtl.pre_trained = 1
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')


def _placeholders():
    for name in ('cv2', 'libtiff', 'h5py', 'openpyxl', 'prefetch_generator'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    # names bound by `from X import Y` at module import time; deliberately inert
    sys.modules['libtiff'].TIFF = None
    sys.modules['openpyxl'].Workbook = None
    sys.modules['openpyxl'].load_workbook = None
    sys.modules['prefetch_generator'].BackgroundGenerator = None


def main():
    assert os.path.isdir(REF), 'reference not present: goldens can only be generated in the build container'
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(REPO, 'oracle'), REF, REPO, os.path.join(REPO, 'dual-modal-fusion_amd', 'dmf')]
    _placeholders()

    from function import function as rf            # reference
    from train import dataset as rds               # reference
    from indicators import kappa as rk             # reference
    from image_convert import IHS as rihs          # reference
    from train.loss_function import qua_loss as r_qua_loss   # reference
    from utils import utils as ru                  # reference
    import synth                                   # product-side scene generator (data only)

    rng = np.random.default_rng(7)

    # ---- G1
    cube = rng.integers(0, 1000, size=(6, 7, 5)).astype(np.float64)
    np.savez(os.path.join(OUT, 'g1_to_tensor.npz'), cube=cube, out=rf.to_tensor(cube))

    # ---- G2
    lab = rng.integers(0, 4, size=(7, 9)).astype(np.uint8)
    cfg2 = {'DATA_DICT': {'t': {'size': [7, 9, 5]}}, 'data_city': 't'}
    the_matrix, matrix_ = rf.split_data_old(lab, cfg2)
    np.savez(os.path.join(OUT, 'g2_split.npz'), label=lab, x=the_matrix[0], y=the_matrix[1], l=the_matrix[2],
             idx0=np.array(matrix_[0]), idx1=np.array(matrix_[1]))

    # ---- G2b  `data_new: 1`: split_data on fixed train / test masks (function.py:172-194).  Drawn from a generator of its
    #           own so that the stream behind G3.. is unchanged.
    rng2 = np.random.default_rng(70)
    tr_mask = (rng2.random((7, 9)) < 0.3).astype(np.uint8) * lab.clip(0, 1)
    te_mask = (rng2.random((7, 9)) < 0.5).astype(np.uint8) * lab.clip(0, 1)       # overlaps the train mask on purpose
    m_new, idx_new = rf.split_data(tr_mask, te_mask, lab, cfg2)
    np.savez(os.path.join(OUT, 'g2b_split_new.npz'), label=lab, train=tr_mask, test=te_mask, x=m_new[0], y=m_new[1], l=m_new[2],
             idx0=np.array(idx_new[0]), idx1=np.array(idx_new[1]), idx2=np.array(idx_new[2]))

    # ---- G3   (reference geometry: PAN at 4x, dataset.py:166,173-176)
    p = 3
    MS = rng.random((7 + p - 1, 9 + p - 1, 5))
    PAN = rng.random((4 * (7 + p - 1), 4 * (9 + p - 1)))
    ds = rds.dataset_dual(MS, PAN, the_matrix, {'patch_size': p})
    items = {}
    for n, i in enumerate((0, 10, 62)):
        ms, pan, l, x, y = ds[i]
        assert isinstance(x, int) and isinstance(y, int) and l.dtype == torch.float32 and l.dim() == 0
        items.update({f'ms{n}': ms.numpy(), f'pan{n}': pan.numpy(), f'l{n}': l.numpy(), f'xy{n}': np.array([x, y])})
    np.savez(os.path.join(OUT, 'g3_dataset.npz'), MS=MS, PAN=PAN, idx=np.array([0, 10, 62]), length=len(ds),
             patch=p, **items)

    # ---- G5   (seed recipe SURVEY appendix A.7)
    torch.manual_seed(0)
    x = torch.randn(40, 17)
    t = torch.randint(1, 17, (10,)).float()
    logits = x[:10].clone().requires_grad_(True)
    ce = ru.make_loss('Criterion', {})
    loss = ce(logits, t.long())
    loss.backward()
    w = torch.nn.Parameter(torch.randn(33))
    w0 = w.detach().clone()
    opt = ru.make_optimizer({'schedule': {'optimizer': 'ADAM', 'lr': 1e-3}}, [w])
    g = torch.randn(33)
    traj = []
    for _ in range(3):
        w.grad = g.clone()
        opt.step()
        traj.append(w.detach().clone().numpy())
        g = g * 0.5 + 0.1
    sch_cfg = {'schedule': {'scheduler': 'ExponentialLR', 'if_scheduler': 1, 'lr': 1e-3, 'base_lr': 5e-4}, 'epoch': 5}
    opt2 = ru.make_optimizer({'schedule': {'optimizer': 'ADAM', 'lr': 1e-3}}, [torch.nn.Parameter(torch.zeros(1))])
    sch = ru.make_scheduler(opt2, sch_cfg)
    lrs = []
    for _ in range(4):
        opt2.step()
        sch.step()
        lrs.append(opt2.param_groups[0]['lr'])
    np.savez(os.path.join(OUT, 'g5_ce_adam.npz'), logits=x[:10].numpy(), target=t.numpy(), ce=loss.item(),
             ce_grad=logits.grad.numpy(), w0=w0.numpy(), g0=torch.randn(0).numpy(), adam_traj=np.stack(traj),
             adam_defaults=json.dumps({k: (list(v) if isinstance(v, tuple) else v)
                                       for k, v in opt.defaults.items() if isinstance(v, (int, float, tuple, bool))}),
             exp_lrs=np.array(lrs))
    # the gradient sequence used above, regenerated deterministically by the test:
    torch.manual_seed(0); torch.randn(40, 17); torch.randint(1, 17, (10,)); torch.randn(33)
    g_first = torch.randn(33)
    np.save(os.path.join(OUT, 'g5_adam_g0.npy'), g_first.numpy())

    # ---- G6
    mats = {
        'm3': [[5, 0, 0], [1, 8, 2], [0, 1, 9]],
        'm5': rng.integers(0, 30, size=(5, 5)).tolist(),
    }
    g6 = {}
    for k, m in mats.items():
        m_np = np.array(m, dtype=np.float64)
        aa, oa, kp, disp = rk.aa_oa(m_np)
        g6[k] = {'matrix': m, 'kappa': float(rk.kappa(m_np)), 'aa': float(aa), 'oa': float(oa), 'kappa_aa_oa': float(kp),
                 'display': [[float(v) for v in row] for row in disp]}
    json.dump(g6, open(os.path.join(OUT, 'g6_kappa.json'), 'w'), indent=1)

    # ---- G7
    pan = np.arange(256, dtype=np.float64).reshape(16, 16)
    un2 = rihs.unsampling(pan, 2)
    p2m = rihs.pan2ms(pan, [4, 4, 4])
    pan_r = rng.random((12, 20))
    p2m_r = rihs.pan2ms(pan_r, [3, 5, 4])
    import random as _random
    _random.seed(5)
    MSr, PANr = rng.random((3, 4, 4)), rng.random((12, 16))
    ihs = rihs.IHS_tran(MSr, PANr)
    np.savez(os.path.join(OUT, 'g7_ihs.npz'), pan=pan, un2=un2, p2m=p2m, pan_r=pan_r, p2m_r=p2m_r,
             ihs_ms=MSr, ihs_pan=PANr, ihs_out=ihs)

    # ---- G8
    torch.manual_seed(0)
    x = torch.randn(40, 17)
    t = torch.randint(1, 17, (10,)).float()
    qcfg = {'device': 'cpu', 'dqtl': {'alpha': 0.1, 'beta': 0.05, 'gamma': 1.0, 'epsilon': 1e-8, 'tao': 0.1}}
    xin = x.clone().requires_grad_(True)
    ql = r_qua_loss()(xin, 10, t, qcfg)
    ql.backward()
    np.savez(os.path.join(OUT, 'g8_qua_loss.npz'), logits=x.numpy(), target=t.numpy(), loss=ql.item(),
             grad=xin.grad.numpy(), cfg=json.dumps(qcfg['dqtl']))

    # ---- G9 (+G4): the real reference Solver.train / Solver.test around the oracle Net
    _trajectory(rf, rds, rk, synth)
    # ---- G10: the real reference toStageSolver.train / .test (stage 2) around the oracle Net
    _trajectory_stage2(rf, rds, rk, rihs, synth)
    print('goldens written to', OUT)


def _trajectory(rf, rds, rk, synth):
    from solver import mainsolver as rms           # reference
    from solver.basesolver import BaseSolver as RBase
    import model.gmfnet as plug                    # oracle/model/gmfnet.py
    from oracle import datapath_ref as dref

    H, W, C, P, S, ncls = 20, 20, 8, 5, 4, 4
    primary, aux, label = synth.make_scene(H, W, C, 1, S, n_classes=ncls, seed=3)
    cfg = {
        'task': 'classification', 'nohup': 0, 'model_name': 'gmfnet', 'time': 1, 'index': 0, 'epoch': 60,
        'device': 'cpu', 'gpu_mode': False, 'patch_size': P, 'Categories_Number': ncls + 1,
        'batchsize': 32, 'test_batchsize': 64, 'color_batchsize': 64, 'train_rate': 0.4, 'verify_rate': 0.1,
        'data_new': 0, 'data_city': 'syn', 'DATA_DICT': {'syn': {'size': [H, W, C], 'color': synth.class_colors(ncls + 1)}},
        'schedule': {'loss': 'Criterion', 'optimizer': 'ADAM', 'if_scheduler': 0, 'scheduler': 'ExponentialLR',
                     'activate': 'Relu', 'lr': 3e-3, 'base_lr': 5e-4},
        'train': {'index': 1, 'pretrained': 0, 'save_best': True}, 'test': {'index': 1, 'save_matrix': 1},
        'color': {'index': 0, 'supervised': 1, 'unsupervised': 1},
        'scale': S, 'aux_bands': 1, 'gmf': {'width': 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0},
    }
    tmp = tempfile.mkdtemp(prefix='dmf_g9_')
    cfg['RESULT_output'] = os.path.join(tmp, 'out') + '/'
    cfg['RESULT_excel'] = os.path.join(tmp, 'r.xlsx')
    os.makedirs(cfg['RESULT_output'])
    cwd = os.getcwd()
    os.chdir(tmp)                                   # Solver.test writes '<time>pan.jpg' into cwd (mainsolver.py:136)
    try:
        torch.manual_seed(3407)                     # test.py:8
        s = rms.Solver.__new__(rms.Solver)
        # -- what BaseSolver.__init__ (basesolver.py:9-61) sets, minus read_tif/data_padding (libtiff/cv2 absent):
        s.cfg, s.task, s.TIME, s.time, s.EPOCH, s.epoch, s.DEVICE = cfg, cfg['task'], cfg['time'], cfg['index'], cfg['epoch'], 0, 'cpu'
        s.num_workers = 0
        s.MS = dref.data_padding(primary, P, S)     # restated padding (parity unpinned: cv2 absent)
        s.PAN = dref.data_padding(aux, P, S)
        xyl, s.matrix_ = rf.split_data_old(label, cfg)          # REAL reference
        order = []

        class Rec(rds.dataset_dual):                             # REAL reference dataset, index-recording
            def __getitem__(self, i):
                order.append(int(i))
                return super().__getitem__(i)

        s.dataset = Rec(s.MS, s.PAN, xyl, cfg)
        s.records = {}
        # -- what Solver.__init__ (mainsolver.py:12-18) sets:
        s.model = s.cur_model = None
        s.train_time = s.test_time = 0
        s.matrix = None

        losses = []
        real_make_loss = rms.make_loss

        def rec_make_loss(kind, c):
            inner = real_make_loss(kind, c)

            class RecLoss(torch.nn.Module):
                def forward(self, out, tgt):
                    v = inner(out, tgt)
                    losses.append(float(v.item()))
                    return v
            return RecLoss()

        rms.make_loss = rec_make_loss
        plug.TRACE.update(enabled=True, init_state=None, logits=[], train_flags=[])
        RBase.dataloader(s)                                       # REAL reference split (basesolver.py:86-105)
        split = {k: np.array(getattr(s, k).dataset.indices) for k in ('train_loader', 'test_loader', 'valid_loader')}
        base = np.array(s.matrix_[1])
        n_before = len(order)
        rms.Solver.train(s)                                       # REAL reference loop (mainsolver.py:40-88)
        train_order = np.array(order[n_before:])
        n_losses_train = len(losses)
        flags = list(plug.TRACE['train_flags'])
        best_state = torch.load(cfg['RESULT_output'] + '0_weights.pth')
        cur = torch.load(cfg['RESULT_output'] + '0_curweights.pth')
        assert set(cur.keys()) == {'state_dict', 'optimizer'}
        n_logits_train = len(plug.TRACE['logits'])
        try:
            rms.Solver.test(s)                                    # REAL reference eval (mainsolver.py:90-148)
        except TypeError:
            pass        # expo_result -> Workbook placeholder; test_matrix is already set (mainsolver.py:147-148)
        test_logits = plug.TRACE['logits'][n_logits_train].numpy()
        aa, oa, kp, _ = rk.aa_oa(s.test_matrix)
        out = dict(
            primary=primary, aux=aux, label=label, cfg=json.dumps({k: v for k, v in cfg.items() if k not in ('RESULT_output', 'RESULT_excel')}),
            labelled=base, split_train=split['train_loader'], split_test=split['test_loader'], split_valid=split['valid_loader'],
            visit_order=train_order, losses=np.array(losses[:n_losses_train]), is_train_call=np.array(flags[:n_logits_train]),
            test_logits=test_logits, test_matrix=s.test_matrix, kappa=kp, aa=aa, oa=oa,
        )
        for k, v in plug.TRACE['init_state'].items():
            out['init.' + k] = v.numpy()
        for k, v in best_state.items():
            out['best.' + k] = v.numpy()
        for k, v in cur['state_dict'].items():
            out['last.' + k] = v.numpy()
        np.savez_compressed(os.path.join(OUT, 'g9_trajectory.npz'), **out)
        print('G9: %d loss calls, kappa %.6f, test n=%d' % (len(losses), kp, int(s.test_matrix.sum())))
    finally:
        rms.make_loss = real_make_loss
        plug.TRACE['enabled'] = False
        os.chdir(cwd)


def _trajectory_stage2(rf, rds, rk, rihs, synth):
    for name, attrs in (('model.generator', {'Generator': None}), ('model.discriminator', {'Discriminator2': None}),
                        ('torchvision', {}), ('torchvision.utils', {'save_image': None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    from solver import tostagesolver as rts        # reference
    from solver import mainsolver as rms           # reference
    from solver.basesolver import BaseSolver as RBase
    import model.gmfnet as plug                    # oracle/model/gmfnet.py
    from oracle import datapath_ref as dref

    H, W, C, P, ncls = 20, 20, 4, 5, 4
    primary, aux, label = synth.make_scene(H, W, C, 1, 4, n_classes=ncls, seed=5)      # 4-band MS + PAN at 4x
    pan4 = rihs.pan2ms(np.asarray(aux, dtype=np.float64), [H, W, 4])                   # REAL reference (IHS.py:14-19)
    g = np.random.default_rng(11)
    ms_gan = primary + 0.15 * g.standard_normal(primary.shape)                         # stand-ins for stage-1 output
    pan_gan = pan4 + 0.15 * g.standard_normal(pan4.shape)
    cfg = {
        'task': 'classification', 'nohup': 0, 'model_name': 'gmfnet', 'time': 1, 'index': 0, 'epoch': 40,
        'device': 'cpu', 'gpu_mode': False, 'patch_size': P, 'Categories_Number': ncls + 1,
        'batchsize': 24, 'test_batchsize': 50, 'color_batchsize': 64, 'train_rate': 0.4, 'verify_rate': 0.1,
        'data_new': 0, 'data_city': 'syn', 'DATA_DICT': {'syn': {'size': [H, W, C], 'color': synth.class_colors(ncls + 1)}},
        'schedule': {'loss': 'qua_loss', 'optimizer': 'ADAM', 'if_scheduler': 0, 'scheduler': 'ExponentialLR',
                     'activate': 'Relu', 'lr': 3e-3, 'base_lr': 5e-4},
        'train': {'index': 1, 'pretrained': 0, 'save_best': True}, 'test': {'index': 1, 'save_matrix': 1},
        'color': {'index': 0, 'supervised': 1, 'unsupervised': 1},
        'dqtl': {'alpha': 0.1, 'beta': 0.05, 'gamma': 1.0, 'epsilon': 1e-8, 'tao': 0.1, 'pre_trained': 1},
        'gmf': {'width': 40, 'hidden': 64, 'pool_sigma': 2.5, 'attention': 0, 'single_input': 1},
    }
    tmp = tempfile.mkdtemp(prefix='dmf_g10_')
    cfg['RESULT_output'] = os.path.join(tmp, 'out') + '/'
    cfg['RESULT_excel'] = os.path.join(tmp, 'r.xlsx')
    os.makedirs(cfg['RESULT_output'])
    cwd = os.getcwd()
    os.chdir(tmp)
    real_make_loss = rms.make_loss
    try:
        torch.manual_seed(3407)
        s = rts.toStageSolver.__new__(rts.toStageSolver)
        s.cfg, s.task, s.TIME, s.time, s.EPOCH, s.epoch, s.DEVICE = cfg, cfg['task'], cfg['time'], cfg['index'], cfg['epoch'], 0, 'cpu'
        s.num_workers = 0
        # what train_stage2 (tostagesolver.py:240-257) builds, with the restated padding (cv2 absent)
        scenes = [dref.data_padding(x, P) for x in (primary, pan4, ms_gan, pan_gan)]
        xyl, s.matrix_ = rf.split_data_old(label, cfg)                       # REAL reference
        order = []

        class Rec(rds.dataset_qua_dqtl):                                     # REAL reference dataset
            def __getitem__(self, i):
                order.append(int(i))
                return super().__getitem__(i)

        s.dataset = Rec(scenes[0], scenes[1], scenes[2], scenes[3], xyl, cfg)
        s.records = {}
        s.model = s.cur_model = None
        s.train_time = s.test_time = 0
        s.matrix = None
        losses = []

        def rec_make_loss(kind, c):
            inner = real_make_loss(kind, c)                                  # REAL reference qua_loss

            class RecLoss(torch.nn.Module):
                def forward(self, out, bs, t, cc):
                    v = inner(out, bs, t, cc)
                    losses.append(float(v.item()))
                    return v
            return RecLoss()

        rms.make_loss = rec_make_loss              # Solver.init_model resolves make_loss in mainsolver's namespace
        plug.TRACE.update(enabled=True, init_state=None, logits=[], train_flags=[])
        RBase.dataloader(s)
        split = {k: np.array(getattr(s, k).dataset.indices) for k in ('train_loader', 'test_loader', 'valid_loader')}
        base = np.array(s.matrix_[1])
        n_before = len(order)
        rts.toStageSolver.train(s)                 # REAL reference loop (tostagesolver.py:259-313)
        train_order = np.array(order[n_before:])
        n_losses_train = len(losses)
        flags = list(plug.TRACE['train_flags'])
        best_state = torch.load(cfg['RESULT_output'] + '0_weights.pth')
        cur = torch.load(cfg['RESULT_output'] + '0_curweights.pth')
        n_logits_train = len(plug.TRACE['logits'])
        try:
            rts.toStageSolver.test(s)              # REAL reference eval (tostagesolver.py:315-346)
        except TypeError:
            pass        # indicator -> expo_result -> Workbook placeholder; test_matrix is already set (:345)
        test_logits = plug.TRACE['logits'][n_logits_train].numpy()
        aa, oa, kp, _ = rk.aa_oa(s.test_matrix)
        out = dict(
            primary=primary, aux=aux, pan4=pan4, ms_gan=ms_gan, pan_gan=pan_gan, label=label,
            cfg=json.dumps({k: v for k, v in cfg.items() if k not in ('RESULT_output', 'RESULT_excel')}),
            labelled=base, split_train=split['train_loader'], split_test=split['test_loader'], split_valid=split['valid_loader'],
            visit_order=train_order, losses=np.array(losses[:n_losses_train]), is_train_call=np.array(flags[:n_logits_train]),
            test_logits=test_logits, test_matrix=s.test_matrix, kappa=kp, aa=aa, oa=oa,
        )
        for k, v in plug.TRACE['init_state'].items():
            out['init.' + k] = v.numpy()
        for k, v in best_state.items():
            out['best.' + k] = v.numpy()
        for k, v in cur['state_dict'].items():
            out['last.' + k] = v.numpy()
        np.savez_compressed(os.path.join(OUT, 'g10_stage2.npz'), **out)
        print('G10: %d loss calls, kappa %.6f, test n=%d' % (len(losses), kp, int(s.test_matrix.sum())))
    finally:
        rms.make_loss = real_make_loss
        plug.TRACE['enabled'] = False
        os.chdir(cwd)


if __name__ == '__main__':
    main()
