"""solver.mainsolver — `Solver(cfg).run()`: train / test / colour (mirror of the reference Solver).

Reference behaviour kept (solver/mainsolver.py:30-209): the network is loaded by name from `model.<model_name>`;
per epoch one pass over the shuffled train loader with CE + ADAM, then the validation pass that stops as soon as
its running loss exceeds the best one (:62-76), best weights -> `<time>_weights.pth` (raw state_dict), every
epoch -> `<time>_curweights.pth` ({'state_dict','optimizer'}); `test()` fills `test_matrix[pred][target]` and calls
`indicator()`; `color()` writes `<time>_pic_1.png` / `_pic_2.png`.

Two execution paths, same arithmetic:
  * fast (default on a GPU): resident scene + epoch plan + fused HIP step (dmf/engine.py) — no patch
    materialisation, no per-step host sync, confusion matrix and label maps built on the device;
  * drop-in (`fast_path: 0`): the reference's own loop body (mainsolver.py:49-55) over materialised batches through
    `Net.forward` / autograd / torch ADAM.
Data parallel (`test.py` under `torch.distributed.run`, fast path only): every rank holds the scene, iterates the SAME
shuffled index stream (same seed) and trains on its contiguous shard of each global batch (a batch that the world size
does not divide is trimmed to the largest multiple); the gradient exchange is the engine's; validation runs on every
rank (identical decisions), test / colour shard the pixels (dmf/parallel.py); rank 0 writes the artefacts.
Deliberate differences: `test()` evaluates the whole test split when `test.full: 1` (the reference always stops
after the first batch, :142 — that stays the default); the t-SNE plot inside `test()` and the visualisation
helpers (:110-136,211-441) are out of scope; `nohup: 1` does not crash (reference bug at :76).
"""
import importlib
import time

import numpy as np
import torch
from PIL import Image
from tqdm import tqdm

from solver.basesolver import BaseSolver
from utils.utils import epoch_hparams, export_optimizer, make_loss, optim_hparams, make_optimizer, make_scheduler, save_checkpoint


class Solver(BaseSolver):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.model = None
        self.cur_model = None
        self.train_time = 0
        self.test_time = 0
        self.matrix = None
        self.engine = None
        self.process_group = None                      # set by the launcher for data-parallel runs (test.py)
        self.comm = None
        self.rank, self.world = 0, 1
        if self.cfg['train']['pretrained']:
            self.init_model()

    def init_model(self):
        lib = importlib.import_module('model.' + self.cfg['model_name'].lower())
        self.model = lib.Net(args=self.cfg)
        self.optimizer = make_optimizer(self.cfg, self.model.parameters())
        self.loss = make_loss(self.cfg['schedule']['loss'], self.cfg)
        self.scheduler = make_scheduler(self.optimizer, self.cfg)

    # ------------------------------------------------------------------ helpers of the fast path
    def _bar(self, it):
        return it if self.cfg['nohup'] else tqdm(it, leave=True)

    @staticmethod
    def _xy_labels(batch):
        x, y, label, _ = batch
        return torch.stack([x, y], 1).to(torch.int32), label.to(torch.int32)

    def _export_optimizer(self):
        """The configured optimiser (ADAM, SGD or RMSprop) with its own state keys, from the engine's flat state vectors
        (checkpoint interchange with the reference's save_checkpoint / load_checkpoint, utils/utils.py:82-102)."""
        eng = self.engine
        group = {'lr': eng.lr}
        if eng.optim == 'ADAM':
            group['betas'] = (eng.b1, eng.b2)
        elif eng.optim == 'SGD':
            group['momentum'] = eng.momentum
        return export_optimizer(self.cfg, self.cur_model.parameters(), self.cur_model._named(), self.cur_model._offsets, eng.m, eng.v,
                                eng.step_count, group)

    # ------------------------------------------------------------------ train
    def train(self):
        time1 = time.time()
        save_best = self.cfg['train']['save_best']
        best_loss = float('inf') if save_best else None
        best_epoch = 0 if save_best else None
        if not self.cfg['train']['pretrained']:
            self.init_model()
        self.cur_model = self.model.to(self.DEVICE)
        if self.fast:
            self._make_engines()
        self.step_losses = []
        while self.epoch < self.EPOCH:
            self.cur_model.train()
            if self.fast:
                last = self._train_epoch_fast()
            else:
                last = self._train_epoch_dropin()
            if save_best:
                self.cur_model.eval()
                val_loss = self._valid_pass(best_loss)
                if val_loss < best_loss:
                    best_loss, best_epoch = val_loss, self.epoch
                    if self.rank == 0:
                        torch.save(self.cur_model.state_dict(), self.cfg['RESULT_output'] + str(self.time) + '_weights.pth')
                    if self.cfg['nohup']:
                        print("best epoch now is {}".format(self.epoch))
            # `<t>_curweights.pth` (model + optimiser, mainsolver.py:83-84) is what an interrupted run resumes from; the reference
            # writes it after EVERY epoch, which on the fast path is half of a small epoch's wall time (tools/solver_epoch_profile.sh:
            # 2.8 of 5.6 ms).  train.save_every: N (NEW, default 1 = the reference) writes it every N-th epoch and after the last.
            every = int(self.cfg['train'].get('save_every', 1) or 1)
            if self.rank == 0 and ((self.epoch + 1) % every == 0 or self.epoch + 1 == self.EPOCH):
                opt = self._export_optimizer() if self.fast else self.optimizer
                save_checkpoint(self.cur_model, opt, self.cfg['RESULT_output'] + str(self.time) + '_curweights.pth')
            if self.world > 1:
                import torch.distributed as dist
                dist.barrier(self.process_group)               # the files exist before any rank goes on to load them
            if self.cfg['nohup']:
                print("{} times {}th epoch is trained, loss {:.6f}".format(self.time, self.epoch, last))
            self.epoch += 1
        self.train_time = time.time() - time1
        self.epoch = 0

    def _make_engines(self):
        from dmf.engine import EvalEngine, TrainEngine
        if self.cfg['schedule']['loss'] != 'Criterion':
            raise ValueError('the fused HIP step implements the Criterion (cross-entropy) loss')
        hp = optim_hparams(self.cfg)                             # ADAM (fused), SGD or RMSprop (utils/utils.py:10-16)
        if self.cfg['batchsize'] % self.world:
            raise ValueError('batchsize %d is not divisible by the %d ranks' % (self.cfg['batchsize'], self.world))
        self.engine = TrainEngine(self.cur_model, self.scene, self.cfg['batchsize'] // self.world, lr=hp['lr'], betas=hp['betas'],
                                  eps=hp['eps'], process_group=self.process_group,
                                  comm=self.comm if hp['optimizer'] == 'ADAM' else None, optimizer=hp['optimizer'],
                                  momentum=hp.get('momentum', 0.0), alpha=hp.get('alpha', 0.99))
        self.eval_engine = EvalEngine(self.cur_model, self.scene, self._eval_chunk())

    def _eval_chunk(self):
        """Pixels per evaluation launch of the fast path.  `test_batchsize` / `color_batchsize` size the reference's host-fed
        loader (mainsolver.py:90-101,155-163); with the scene resident every pixel is independent of its batch, and a launch of
        256 patches is bound by the host (0.13 of the HBM roof on a 512x512x224 scene) where 16,384 reach 0.62
        (tools/eval_bench.py; identical class maps)."""
        big = 4096 if self.cur_model.arch['attention'] else 16384
        return max(self.cfg['test_batchsize'], self.cfg['color_batchsize'], big)

    def _train_epoch_fast(self):
        eng, B = self.engine, self.cfg['batchsize'] // self.world
        hp = epoch_hparams(self.cfg, self.epoch)              # lr (and, under OneCycleLR, beta1 / momentum) of this epoch
        eng.lr = float(hp['lr'])
        if 'betas' in hp:
            eng.b1, eng.b2 = float(hp['betas'][0]), float(hp['betas'][1])
        if eng.optim == 'SGD':
            eng.momentum = float(hp['momentum'])
        batches = [self._xy_labels(b) for b in self.train_index_loader]      # the epoch's shuffled coordinates
        if self.world > 1:                                                   # this rank's contiguous shard of every batch
            cut = []
            for xy, lab in batches:
                per = xy.shape[0] // self.world                             # (a remainder is dropped, see the module text)
                if per:
                    cut.append((xy[self.rank * per:(self.rank + 1) * per], lab[self.rank * per:(self.rank + 1) * per]))
            batches = cut
        full = [b for b in batches if b[0].shape[0] == B]
        losses = []
        if full:
            eng.load_plan(torch.cat([b[0] for b in full]), torch.cat([b[1] for b in full]))
            # steps_per_graph: N > 0 replays captured hipGraphs of N steps, 0 launches step by step from Python, -1 (default)
            # hands the whole epoch to the library's launch loop (dmf_train_plan_steps) where that exists, else step by step
            eng.run_plan(len(full), int(self.cfg.get('steps_per_graph', -1)))
            losses = eng.mean_losses().tolist()
        for xy, lab in batches:
            if xy.shape[0] != B:                                             # DataLoader keeps the short last batch
                eng.step(xy.to(self.DEVICE), lab.to(self.DEVICE))
                losses.append(float(eng.loss[:xy.shape[0]].mean().item()))
        self._check_exchange()
        self.step_losses += losses
        return losses[-1] if losses else float('nan')

    def _check_exchange(self):
        """A timed-out wait of the one-shot gradient exchange leaves the ranks with different weights (the kernel sums what
        it has and goes on).  Every rank learns of it here, once per epoch and before anything is saved: all ranks stop."""
        if self.comm is None:
            return
        import torch.distributed as dist
        bad = torch.tensor([int(self.comm.status())], dtype=torch.int32,
                           device=self.DEVICE if dist.get_backend(self.process_group) == 'nccl' else 'cpu')
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.process_group)
        if int(bad.item()) != 0:
            raise RuntimeError('epoch %d: a gradient exchange timed out on at least one rank; the replicas have diverged. '
                               'Restart from the last checkpoint with xgmi_exchange: 0 (RCCL all-reduce).' % self.epoch)

    def _train_epoch_dropin(self):
        loader = self._bar(self.train_loader)
        last = float('nan')
        for data1, data2, target, _, _ in loader:
            data1, data2, target = data1.to(self.DEVICE), data2.to(self.DEVICE), target.to(self.DEVICE)
            self.optimizer.zero_grad()
            output = self.cur_model(data1, data2)
            loss = self.loss(output, target.long())
            loss.backward()
            self.optimizer.step()
            last = loss.item()
            self.step_losses.append(last)
            if not self.cfg['nohup']:
                loader.set_postfix(ls=last, ep=self.epoch, tm=self.time, m='train', d=self.cfg['device'])
        if self.cfg['schedule']['if_scheduler']:
            self.scheduler.step()
        return last

    def _forward_batch(self, batch):
        """(logits, target int64 on device, x, y) for a batch of either loader twin."""
        if self.fast:
            xy, lab = self._xy_labels(batch)
            logits, _ = self.eval_engine.predict(xy.to(self.DEVICE))
            return logits, lab.to(self.DEVICE).long(), batch[0], batch[1]
        data1, data2, target, x, y = batch
        return self.cur_model(data1.to(self.DEVICE), data2.to(self.DEVICE)), target.to(self.DEVICE).long(), x, y

    def _valid_pass(self, best_loss):
        ce = torch.nn.CrossEntropyLoss()
        val_loss = 0.0
        if self.fast:
            # The reference adds `loss.item() * n` per batch and stops once the sum passes best_loss (mainsolver.py:65-75):
            # one host sync per batch.  Here the same double-precision sum stays on the device and is read once; the terms
            # are non-negative, so "the full sum is below best_loss" decides exactly what the early exit decides.
            tot = torch.zeros((), dtype=torch.float64, device=self.DEVICE)
            with torch.no_grad():
                for batch in self.valid_index_loader:
                    xy, lab = self._xy_labels(batch)
                    # the evaluation launch's own per-patch cross-entropy (dmf_forward_ce) where the shape has it ...
                    part = self.eval_engine.ce_sum(xy.to(self.DEVICE), lab.to(self.DEVICE)) if hasattr(self.eval_engine, 'ce_sum') else None
                    if part is None:                         # ... else torch's on the logits (attention network)
                        logits, target, _, _ = self._forward_batch(batch)
                        part = ce(logits, target).double() * target.shape[0]
                    tot += part
            return float(tot.item())
        with torch.no_grad():
            for batch in (self.valid_index_loader if self.fast else self.valid_loader):
                logits, target, _, _ = self._forward_batch(batch)
                val_loss += ce(logits, target).item() * target.shape[0]
                if val_loss > best_loss:
                    break
        return val_loss

    # ------------------------------------------------------------------ test
    def _load_weights(self, best):
        path = self.cfg['RESULT_output'] + str(self.time) + ('_weights.pth' if best else '_curweights.pth')
        sd = torch.load(path, map_location=self.DEVICE, weights_only=True)
        self.cur_model.load_state_dict(sd if best else sd['state_dict'])

    def _ensure_model(self):
        if self.cur_model is None:
            self.init_model()
            self.cur_model = self.model.to(self.DEVICE)
        if self.fast and getattr(self, 'eval_engine', None) is None:
            self._make_eval_engine()

    def _make_eval_engine(self):
        from dmf.engine import EvalEngine
        self.eval_engine = EvalEngine(self.cur_model, self.scene, self._eval_chunk())

    def test(self):
        time1 = time.time()
        self._ensure_model()
        self._load_weights(self.cfg['train']['save_best'])
        self.cur_model.eval()
        K = self.cfg['Categories_Number']
        full = bool(self.cfg['test'].get('full', 0))
        with torch.no_grad():
            if self.fast and full:
                # whole split in evaluation chunks of the engine's own size (data parallel: every rank classifies its part
                # of the pixels, the matrices are summed)
                parts = [self._xy_labels(b) for b in self.test_index_loader]
                matrix = self.eval_engine.confusion(torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]),
                                                    process_group=self.process_group)
                test_matrix = matrix.cpu().numpy().astype(np.float64)
            elif self.fast:
                from dmf import lib
                matrix = torch.zeros(K, K, dtype=torch.int64, device=self.DEVICE)
                for batch in self.test_index_loader:
                    xy, lab = self._xy_labels(batch)
                    _, pred = self.eval_engine.predict(xy.to(self.DEVICE))
                    lib.confusion_accum(pred, lab.to(self.DEVICE), K, matrix)
                    if not full:
                        break                                                # mainsolver.py:142
                test_matrix = matrix.cpu().numpy().astype(np.float64)
            else:
                test_matrix = np.zeros([K, K])
                for batch in self.test_loader:
                    logits, target, _, _ = self._forward_batch(batch)
                    pred = logits.data.max(1)[1].cpu().numpy()
                    np.add.at(test_matrix, (pred, target.cpu().numpy()), 1)
                    if not full:
                        break
        self.test_time = time.time() - time1
        self.test_matrix = test_matrix
        if self.rank == 0:
            self.indicator()
        else:
            from indicators.kappa import aa_oa_quiet
            self.result = list(aa_oa_quiet(test_matrix)) + [None]

    # ------------------------------------------------------------------ colour
    def color(self):
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
                m = torch.zeros(H, W, dtype=torch.int32, device=self.DEVICE) if self.fast else np.zeros([H, W], dtype=np.int64)
                if use and self.fast:      # all pixels, in evaluation chunks of the engine's own size (_eval_chunk)
                    xy_all = torch.cat([self._xy_labels(b)[0] for b in loaders[0]])
                    m = self.eval_engine.label_map(xy_all, H, W, process_group=self.process_group)
                elif use:
                    for batch in loaders[0 if self.fast else 1]:
                        if self.fast:
                            from dmf import lib
                            xy, _ = self._xy_labels(batch)
                            xy = xy.to(self.DEVICE)
                            _, pred = self.eval_engine.predict(xy)
                            lib.labelmap_write(pred, xy, W, m)
                        else:
                            logits, _, x, y = self._forward_batch(batch)
                            m[np.asarray(x), np.asarray(y)] = logits.data.max(1)[1].cpu().numpy()
                maps.append(m.cpu().numpy() if self.fast else m)
        label_np1 = maps[0]
        label_np2 = np.where(maps[1] != 0, maps[1], maps[0]) if self.cfg['color']['unsupervised'] else maps[0]
        self.label_maps = (label_np1, label_np2)
        if self.cfg['color']['supervised'] and self.rank == 0:
            Image.fromarray(lut[label_np1]).save(self.cfg['RESULT_output'] + str(self.time) + "_pic_1.png")
            Image.fromarray(lut[label_np2]).save(self.cfg['RESULT_output'] + str(self.time) + "_pic_2.png")

    def run(self):
        while self.time < self.TIME:
            self.dataloader()
            if self.cfg['train']['index']:
                self.train()
            if self.cfg['test']['index']:
                self.test()
            if self.cfg['color']['index']:
                self.color()
            self.time += 1
