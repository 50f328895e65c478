"""The CPU oracle (oracle/datapath_ref.py) against the fixtures captured from the real reference
(oracle/make_goldens.py).  CPU-only; bit-exact where the arithmetic is integer or a pure re-ordering,
1e-12 relative otherwise (float64 reductions in a different order)."""
import json
import os

import numpy as np
import torch

from oracle import datapath_ref as d
from oracle.gmfnet_ref import adam_step_ref


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_g1_to_tensor(golden_dir):
    g = _g(golden_dir, 'g1_to_tensor.npz')
    out = d.to_tensor(g['cube'])
    assert np.array_equal(out, g['out'])
    assert out.min() == 0.0 and out.max() == 1.0


def test_padding_shape_and_rule():
    a = np.arange(2 * 3 * 2, dtype=np.float64).reshape(2, 3, 2)
    out = d.data_padding(a, 3)
    assert out.shape == (4, 5, 2)
    n = d.to_tensor(a)
    assert np.array_equal(out[:2, :3], n)                 # top-left anchored: original stays at [0:H, 0:W]
    assert np.array_equal(out[2], out[0]) and np.array_equal(out[3, :3], n[0 - 0, :3] * 0 + out[1, :3] * 0 + out[3, :3])
    assert np.array_equal(out[:, 3], out[:, 1])           # reflect-101: no edge repeat
    pan = np.arange(8 * 8, dtype=np.float64).reshape(8, 8)
    assert d.data_padding(pan, 3).shape == (8 + 12 - 1, 8 + 12 - 1)   # 2-D pads 4*patch-1 (function.py:101)


def test_g2_split(golden_dir):
    g = _g(golden_dir, 'g2_split.npz')
    m, idx = d.split_data_old(g['label'], [7, 9, 5])
    for a, k in zip(m, ('x', 'y', 'l')):
        assert a.dtype == np.float64 and a.shape == (63, 1) and np.array_equal(a, g[k])
    assert idx[0] == g['idx0'].tolist() and idx[1] == g['idx1'].tolist()


def test_g3_dataset(golden_dir):
    g = _g(golden_dir, 'g3_dataset.npz')
    lab = _g(golden_dir, 'g2_split.npz')
    xyl = [lab['x'], lab['y'], lab['l']]
    assert int(g['length']) == 63
    for n, i in enumerate(g['idx']):
        ms, pan, l, x, y = d.dataset_dual_item(g['MS'], g['PAN'], xyl, int(i), int(g['patch']), 4)
        assert ms.dtype == torch.float32 and pan.dtype == torch.float32 and l.dtype == torch.float32 and l.dim() == 0
        assert np.array_equal(ms.numpy(), g[f'ms{n}']) and np.array_equal(pan.numpy(), g[f'pan{n}'])
        assert float(l) == float(g[f'l{n}']) and [x, y] == g[f'xy{n}'].tolist()
        assert isinstance(x, int) and isinstance(y, int)


def test_g5_ce_adam(golden_dir):
    g = _g(golden_dir, 'g5_ce_adam.npz')
    logits = torch.from_numpy(g['logits']).requires_grad_(True)
    loss = d.cross_entropy(logits, torch.from_numpy(g['target']))
    loss.backward()
    assert abs(loss.item() - float(g['ce'])) < 1e-6 and abs(float(g['ce']) - 3.2011947632) < 1e-6
    assert np.allclose(logits.grad.numpy(), g['ce_grad'], atol=1e-7)
    dflt = json.loads(str(g['adam_defaults']))
    assert dflt['lr'] == 1e-3 and dflt['betas'] == [0.9, 0.999] and dflt['eps'] == 1e-8 and dflt['weight_decay'] == 0
    w = torch.from_numpy(g['w0']).clone()
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    gr = torch.from_numpy(np.load(os.path.join(golden_dir, 'g5_adam_g0.npy')))
    for step in range(1, 4):
        adam_step_ref(w, gr, m, v, step)
        assert np.allclose(w.numpy(), g['adam_traj'][step - 1], rtol=0, atol=2e-7)
        gr = gr * 0.5 + 0.1
    assert np.allclose(g['exp_lrs'], [d.exponential_lr(1e-3, 0.98, e) for e in range(1, 5)], rtol=1e-12)


def test_g6_kappa(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'g6_kappa.json')))
    assert abs(g['m3']['kappa'] - 0.7620137299771167) < 1e-15
    for k, e in g.items():
        aa, oa, kp, disp = d.aa_oa(e['matrix'])
        assert abs(d.kappa(e['matrix']) - e['kappa']) < 1e-12 and abs(kp - e['kappa']) < 1e-12
        assert abs(aa - e['aa']) < 1e-12 and abs(oa - e['oa']) < 1e-12
        assert np.allclose(np.array(disp), np.array(e['display']), rtol=1e-12)
    # class-0 quirk: OA keeps the class-0 column in its denominator (kappa.py:82)
    assert abs(g['m3']['oa'] - 17 / 26) < 1e-12


def test_confusion_rows_are_predictions():
    m = d.confusion([1, 1, 2], [1, 2, 2], 3)
    assert m[1][2] == 1 and m[2][1] == 0 and m.sum() == 3


def test_g7_ihs(golden_dir):
    g = _g(golden_dir, 'g7_ihs.npz')
    assert np.array_equal(d.unsampling(g['pan'], 2), g['un2'])
    p2m = d.pan2ms(g['pan'], [4, 4, 4])
    assert np.array_equal(p2m, g['p2m']) and p2m[0, 0].tolist() == [8.5, 40.5, 10.5, 42.5]
    assert np.allclose(d.pan2ms(g['pan_r'], [3, 5, 4]), g['p2m_r'], rtol=1e-15)
    # IHS_tran is algebraically the identity on PAN: the fixture proves it for the reference
    assert np.allclose(g['ihs_out'], g['ihs_pan'], atol=1e-12)


def test_g8_qua_loss(golden_dir):
    g = _g(golden_dir, 'g8_qua_loss.npz')
    c = json.loads(str(g['cfg']))
    x = torch.from_numpy(g['logits']).requires_grad_(True)
    loss = d.qua_loss(x, 10, torch.from_numpy(g['target']), c['alpha'], c['beta'], c['gamma'], c['epsilon'], c['tao'])
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 1e-6 and abs(float(g['loss']) - 0.4733544886) < 1e-6
    assert np.allclose(x.grad.numpy(), g['grad'], atol=1e-7)
    assert abs(np.linalg.norm(g['grad']) - 0.0682623833) < 1e-6


def test_split_sizes():
    assert d.split_sizes(285, 0.4, 0.1) == (114, 143, 28)
