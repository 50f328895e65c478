"""solver.tostagesolver — `toStageSolver(cfg).run()`: the reference's two-stage path (solver/tostagesolver.py:20-414).

Stage 2 is built (tostagesolver.py:240-414): the four co-registered 4-band scenes (ms, pan = `pan.npy`, ms_gan, pan_gan)
are padded, sliced by `dataset_qua_dqtl`, concatenated on the batch axis and pushed through ONE single-input network
(`model.<model_name>.Net(args=cfg)` called as `net(data)`; this build's GMFNet takes the band mean of its input as the
auxiliary modality, cfg['gmf']['single_input'] = 1), trained with `qua_loss` + ADAM, best epoch by the early-stopping
validation loop, prediction = argmax softmax(out[:bs] + out[bs:2bs]).  Two execution paths as in solver.mainsolver:
  * fast (default on a GPU): the four scenes resident as one tall scene, epoch plan, four HIP launches per step
    (dmf.engine.QuaTrainEngine: forward, dmf_qua_loss, backward, reduce+ADAM), on-device confusion matrix / label maps;
  * drop-in (`fast_path: 0`): the reference's loop body (:268-278) through `Net.forward(data)` / autograd / torch ADAM
    with `train.loss_function.qua_loss` (the same HIP loss kernel behind an autograd Function).
Stage 1 (:86-238) trains `model.generator` / `model.discriminator`, which the reference does not ship: it
is NOT built.  Run stage 2 on stage-1 outputs that already exist (`dqtl.pre_trained: 1`: `msgan.npy`, `pangan.npy`
under cfg['expo_result'] + cfg['dqtl']['WEIGHTS'], as the reference does at :241-243).  `pan.npy` (:246) is produced
from the PAN image with `pan2ms` (image_convert/IHS.py:14-19, on the GPU) when the file does not exist.
Deliberate differences: label maps are written as PNG (the reference writes .jpg here and .png in Solver); `nohup: 1`
works (reference bug at :300); the t-SNE / feature visualisation helpers (:416-530) are out of scope.
"""
import os
import time

import numpy as np
import torch
from PIL import Image

from function.function import data_padding, data_show, split_data_old
from solver.mainsolver import Solver
from train.dataset import dataset_qua_dqtl
from utils.utils import epoch_hparams, optim_hparams


class toStageSolver(Solver):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.qua_scene = None
        self.ms_gan = self.pan_gan = None

    # ------------------------------------------------------------------ stage 1 (not built)
    def train_stage1(self):
        raise NotImplementedError('stage 1 (GAN pre-fusion, tostagesolver.py:86-238) needs model.generator / '
                                  'model.discriminator, which the reference does not ship; provide msgan.npy / '
                                  'pangan.npy and set dqtl.pre_trained: 1')

    # ------------------------------------------------------------------ stage 2 data (tostagesolver.py:240-257)
    def train_stage2(self):
        cfg = self.cfg
        d = cfg['dqtl']
        if not d.get('pre_trained'):
            self.train_stage1()
        base = cfg.get('expo_result', '') + d.get('WEIGHTS', '')
        self.ms_gan = np.load(base + 'msgan.npy')
        self.pan_gan = np.load(base + 'pangan.npy')
        pan_path = cfg['data_address'] + '/pan.npy'
        if os.path.exists(pan_path):
            PAN = np.load(pan_path)
        else:
            from image_convert.IHS import pan2ms, pan2ms_gpu
            size = [self.ms.shape[0], self.ms.shape[1], 4]
            PAN = pan2ms_gpu(self.pan, size, self.DEVICE) if str(self.DEVICE).startswith('cuda') else pan2ms(self.pan, size)
        scenes = [data_padding(x, cfg, 'ms') for x in (self.ms, PAN, self.ms_gan, self.pan_gan)]
        label_np = np.load(cfg['data_address'] + 'label.npy')
        data_show(label_np)
        xyl_matrix, self.matrix_ = split_data_old(label_np, cfg)
        self.dataset = dataset_qua_dqtl(scenes[0], scenes[1], scenes[2], scenes[3], xyl_matrix, cfg)
        self.index_dataset = self.dataset.index_view()
        if self.fast:
            from dmf.engine import QuaScene
            self.qua_scene = QuaScene(scenes, self.DEVICE)

    # ------------------------------------------------------------------ hooks of Solver.train
    def _make_engines(self):
        from dmf.engine import QuaTrainEngine
        if self.cfg['schedule']['loss'] != 'qua_loss':
            raise ValueError('stage 2 trains with schedule.loss: qua_loss')
        hp = optim_hparams(self.cfg)
        self.engine = QuaTrainEngine(self.cur_model, self.qua_scene, self.cfg['batchsize'], self.cfg['dqtl'], lr=hp['lr'],
                                     betas=hp['betas'], eps=hp['eps'], optimizer=hp['optimizer'],
                                     momentum=hp.get('momentum', 0.0), alpha=hp.get('alpha', 0.99))
        self._make_eval_engine()

    def _make_eval_engine(self):
        from dmf.engine import QuaEvalEngine
        # (evaluation chunks of the engine's own size, as in Solver._eval_chunk: the configured sizes belong to the host-fed loader)
        self.eval_engine = QuaEvalEngine(self.cur_model, self.qua_scene,
                                         max(self.cfg['test_batchsize'], self.cfg['color_batchsize'], 8192), self.cfg['dqtl'])

    def _train_epoch_fast(self):
        eng, B = self.engine, self.cfg['batchsize']
        hp = epoch_hparams(self.cfg, self.epoch)              # lr (and, under OneCycleLR, beta1 / momentum) of this epoch
        eng.lr = float(hp['lr'])
        if 'betas' in hp:
            eng.b1, eng.b2 = float(hp['betas'][0]), float(hp['betas'][1])
        if eng.optim == 'SGD':
            eng.momentum = float(hp['momentum'])
        batches = [self._xy_labels(b) for b in self.train_index_loader]
        full = [b for b in batches if b[0].shape[0] == B]
        losses = []
        if full:
            eng.load_plan(torch.cat([b[0] for b in full]), torch.cat([b[1] for b in full]))
            eng.run_plan(len(full), int(self.cfg.get('steps_per_graph', 0)) if eng.unit else 0)
            losses = eng.losses().tolist()
        for xy, lab in batches:
            if xy.shape[0] != B:                                             # DataLoader keeps the short last batch
                eng.step(xy, lab)
                losses.append(float(eng.loss.item()))
        self.step_losses += losses
        return losses[-1] if losses else float('nan')

    def _train_epoch_dropin(self):
        loader = self._bar(self.train_loader)
        last = float('nan')
        for data1, data2, data3, data4, target, _, _ in loader:
            data = torch.concat([data1, data2, data3, data4]).to(self.DEVICE)            # tostagesolver.py:270-272
            target = target.to(self.DEVICE)
            bs = len(data1)
            self.optimizer.zero_grad()
            output = self.cur_model(data)
            loss = self.loss(output, bs, target, self.cfg)
            loss.backward()
            self.optimizer.step()
            last = loss.item()
            self.step_losses.append(last)
            if not self.cfg['nohup']:
                loader.set_postfix(loss=last, epoch=self.epoch, time=self.time, mode='train')
        if self.cfg['schedule']['if_scheduler']:
            self.scheduler.step()
        return last

    def _valid_pass(self, best_loss):
        val_loss = 0.0
        with torch.no_grad():
            if self.fast:
                for batch in self.valid_index_loader:
                    xy, lab = self._xy_labels(batch)
                    val_loss += self.eval_engine.loss_value(xy, lab).item() * xy.shape[0]
                    if val_loss > best_loss:
                        break
            else:
                for data1, data2, data3, data4, target, _, _ in self.valid_loader:
                    data = torch.concat([data1, data2, data3, data4]).to(self.DEVICE)
                    output = self.cur_model(data)
                    val_loss += self.loss(output, len(data1), target.to(self.DEVICE), self.cfg).item() * data1.size(0)
                    if val_loss > best_loss:
                        break
        return val_loss

    def _pair_pred(self, batch):
        """(pred int32 [bs] on the device, target, x, y) — tostagesolver.py:337."""
        from dmf import lib
        if self.fast:
            xy, lab = self._xy_labels(batch)
            _, pred = self.eval_engine.predict(xy)
            return pred, lab.to(self.DEVICE), xy
        data1, data2, _, _, target, x, y = batch
        bs = len(data1)
        out = self.cur_model(torch.concat([data1, data2]).to(self.DEVICE))
        pred = torch.empty(bs, dtype=torch.int32, device=out.device)
        lib.pair_argmax(out.contiguous(), bs, pred)
        return pred, target.to(self.DEVICE).to(torch.int32), torch.stack([torch.as_tensor(x), torch.as_tensor(y)], 1).to(torch.int32)

    # ------------------------------------------------------------------ test / colour (tostagesolver.py:315-401)
    def test(self):
        from dmf import lib
        time1 = time.time()
        self._ensure_model()
        self._load_weights(self.cfg['train']['save_best'])
        self.cur_model.eval()
        K = self.cfg['Categories_Number']
        matrix = torch.zeros(K, K, dtype=torch.int64, device=self.DEVICE)
        with torch.no_grad():
            if self.fast:                                    # the whole split in the engine's own chunks
                parts = [self._xy_labels(b) for b in self.test_index_loader]
                matrix = self.eval_engine.confusion(torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), matrix)
            else:
                for batch in self.test_loader:               # every batch (:331-341)
                    pred, target, _ = self._pair_pred(batch)
                    lib.confusion_accum(pred, target.contiguous(), K, matrix)
        self.test_time = time.time() - time1
        self.test_matrix = matrix.cpu().numpy().astype(np.float64)
        self.indicator()

    def color(self):
        from dmf import lib
        self._ensure_model()
        self._load_weights(True)
        self.cur_model.eval()
        size = self.cfg['DATA_DICT'][self.cfg['data_city']]['size']
        H, W = int(size[0]), int(size[1])
        lut = np.asarray(self.cfg['DATA_DICT'][self.cfg['data_city']]['color'], dtype=np.uint8)
        maps = []
        with torch.no_grad():
            for use, loaders in ((self.cfg['color']['supervised'], (self.color_index_loader1, self.color_loader1)),
                                 (self.cfg['color']['unsupervised'], (self.color_index_loader2, self.color_loader2))):
                m = torch.zeros(H, W, dtype=torch.int32, device=self.DEVICE)
                if use and self.fast:
                    m = self.eval_engine.label_map(torch.cat([self._xy_labels(b)[0] for b in loaders[0]]), H, W, m)
                elif use:
                    for batch in loaders[1]:
                        pred, _, xy = self._pair_pred(batch)
                        lib.labelmap_write(pred, xy.to(self.DEVICE).contiguous(), W, m)
                maps.append(m.cpu().numpy())
        label_np1 = maps[0]
        label_np2 = np.where(maps[1] != 0, maps[1], maps[0]) if self.cfg['color']['unsupervised'] else maps[0]
        self.label_maps = (label_np1, label_np2)
        if self.cfg['color']['supervised']:
            Image.fromarray(lut[label_np1]).save(self.cfg['RESULT_output'] + str(self.time) + "_pic_1.png")
            Image.fromarray(lut[label_np2]).save(self.cfg['RESULT_output'] + str(self.time) + "_pic_2.png")

    def run(self):
        self.train_stage2()
        super().run()
