"""solver.basesolver — scene loading, pixel table, splits and loaders (mirror of the reference BaseSolver).

Follows solver/basesolver.py:9-126 of the reference: read `ms4.tif` / `pan.tif`, normalise + pad, load `label.npy`,
build the row-major pixel table, `dataset_dual`, `random_split` of the labelled pixels with the GLOBAL torch RNG into
train / test / valid, five DataLoaders, `indicator()`.  Additions for the GPU path: the padded scenes are also kept
resident in HBM (`self.scene`), and every loader has an index-only twin (`*_index_loader`) that yields pixel
coordinates instead of materialised patches.  Iterating either twin consumes the global RNG exactly like the
reference's loader does, so a seeded run visits the same patches in the same order.
"""
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from function.function import data_padding, data_padding_aux, data_show, label_mat2np, read_tif, split_data, split_data_old
from indicators.kappa import aa_oa, expo_result
from train.dataset import collate_batched, dataset_dual


class BaseSolver:
    def __init__(self, cfg):
        self.cfg = cfg
        self.task = cfg['task']
        self.TIME = cfg['time']
        self.time = cfg['index']
        self.EPOCH = cfg['epoch']
        self.epoch = 0
        self.DEVICE = cfg['device']
        self.timestamp = int(time.time())
        self.num_workers = cfg['threads'] if cfg.get('gpu_mode') else 0

        self.ms = read_tif(cfg, 'ms')
        self.pan = read_tif(cfg, 'pan')
        self.MS = data_padding(self.ms, cfg, 'ms')
        self.PAN = data_padding(self.pan, cfg, 'pan') if self.pan.ndim == 2 else data_padding_aux(self.pan, cfg)

        label_path = cfg['data_address'] + 'label.npy'
        if not os.path.exists(label_path):
            label_mat2np(cfg)                                   # basesolver.py:34-35
        label_np = np.load(label_path)
        data_show(label_np)
        self.label_np = label_np
        self.data_new = cfg.get('data_new') == 1
        if self.data_new:                                       # basesolver.py:28-30,38-40: fixed train / test masks
            self.train_label = np.load(cfg['data_address'] + 'train.npy')
            self.test_label = np.load(cfg['data_address'] + 'test.npy')
            xyl_matrix, self.traintest_index = split_data(self.train_label, self.test_label, label_np, cfg)
            _, self.matrix_ = split_data_old(label_np, cfg)
        else:
            xyl_matrix, self.matrix_ = split_data_old(label_np, cfg)
        self.xyl = xyl_matrix
        if cfg.get('use_h5'):
            raise AttributeError("not finished")          # as the reference (basesolver.py:45-46)
        self.dataset = dataset_dual(self.MS, self.PAN, xyl_matrix, cfg)
        self.index_dataset = self.dataset.index_view()
        print('All dataset size:', len(self.dataset))
        self.records = {'Epoch': [], 'PSNR': [], 'SSIM': [], 'Loss': []}
        self.scene = None
        self.fast = bool(cfg.get('fast_path', 1)) and str(self.DEVICE).startswith('cuda')
        if self.fast:
            from dmf.engine import Scene
            self.scene = Scene(self.MS, self.PAN, self.DEVICE)

    def _loader(self, subset, batch, shuffle):
        twin = Subset(self.index_dataset, indices=subset.indices)
        return (DataLoader(dataset=subset, batch_size=batch, shuffle=shuffle, num_workers=self.num_workers),
                DataLoader(dataset=twin, batch_size=batch, shuffle=shuffle, num_workers=0, collate_fn=collate_batched))

    def dataloader(self):
        cfg = self.cfg
        if self.data_new:
            # basesolver.py:64-84: the whole train mask trains; the test mask is split (global RNG) into test / valid
            train_data = Subset(self.dataset, indices=self.traintest_index[1])
            test_data = Subset(self.dataset, indices=self.traintest_index[2])
            valid_size = int(cfg['verify_rate'] * len(test_data))
            test_size = len(test_data) - valid_size
            test_dataset, valid_dataset = torch.utils.data.random_split(test_data, [test_size, valid_size])
            base = np.asarray(self.traintest_index[2])

            def flat2(s):
                return Subset(self.dataset, indices=base[np.asarray(s.indices)].tolist())

            self.train_loader, self.train_index_loader = self._loader(train_data, cfg['batchsize'], True)
            self.test_loader, self.test_index_loader = self._loader(flat2(test_dataset), cfg['test_batchsize'], False)
            self.valid_loader, self.valid_index_loader = self._loader(flat2(valid_dataset), cfg['color_batchsize'], False)
            self.color_loader1, self.color_index_loader1 = self._loader(Subset(self.dataset, indices=self.matrix_[1]),
                                                                        cfg['test_batchsize'], False)
            self.color_loader2, self.color_index_loader2 = self._loader(Subset(self.dataset, indices=self.matrix_[0]),
                                                                        cfg['test_batchsize'], False)
            return
        train_data = Subset(self.dataset, indices=self.matrix_[1])
        train_size = int(cfg['train_rate'] * len(train_data))
        valid_size = int(cfg['verify_rate'] * len(train_data))
        test_size = len(train_data) - train_size - valid_size
        train_dataset, test_dataset, valid_dataset = torch.utils.data.random_split(
            train_data, [train_size, test_size, valid_size])
        # random_split returns Subsets of `train_data`; flatten to indices into the full dataset
        base = np.asarray(self.matrix_[1])

        def flat(s):
            return Subset(self.dataset, indices=base[np.asarray(s.indices)].tolist())

        self.train_loader, self.train_index_loader = self._loader(flat(train_dataset), cfg['batchsize'], True)
        self.test_loader, self.test_index_loader = self._loader(flat(test_dataset), cfg['test_batchsize'], False)
        self.valid_loader, self.valid_index_loader = self._loader(flat(valid_dataset), cfg['color_batchsize'], False)
        color_data = Subset(self.dataset, indices=self.matrix_[0])
        self.color_loader1, self.color_index_loader1 = self._loader(train_data, cfg['test_batchsize'], False)
        self.color_loader2, self.color_index_loader2 = self._loader(color_data, cfg['test_batchsize'], False)

    def indicator(self):
        if self.cfg['test']['save_matrix']:
            np.save(self.cfg['RESULT_output'] + str(self.time) + "_matrix.npy", self.test_matrix)
        result = aa_oa(self.test_matrix)
        self.result = result
        expo_result(result, self.cfg, [self.train_time, self.test_time], self.time)

    def train(self):
        raise NotImplementedError

    def eval(self):
        raise NotImplementedError

    def run(self):
        while self.time < self.TIME:
            self.train()
            self.time += 1
