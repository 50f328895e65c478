"""train.dataset — the patch API (mirror of the reference module of the same name).

`dataset_dual[i]` returns the reference's tuple `(ms [C,p,p] f32, pan [C2,sp,sp] f32, label 0-dim f32, x:int, y:int)`
(train/dataset.py:158-188) with the aux/primary resolution ratio taken from cfg['scale'] (reference: 4).
`dataset_dual.index_view()` is the same dataset without the patch materialisation: `(x, y, label, index)` as ints —
the resident-scene fast path gathers the pixels on the GPU, so the host only shuffles coordinates.
"""
import numpy as np
import torch
from torch.utils.data import Dataset


class dataset_dual(Dataset):
    def __init__(self, ms, pan, xyl, cfg):
        self.MS = ms
        self.PAN = pan
        self.Label = xyl[2]
        self.x = xyl[0]
        self.y = xyl[1]
        self.scale = int(cfg.get('scale', 4))
        self.ms_size = cfg['patch_size']
        self.pan_size = cfg['patch_size'] * self.scale

    def __getitem__(self, index):
        s = self.scale
        ms_x, ms_y = int(np.asarray(self.x[index]).reshape(-1)[0]), int(np.asarray(self.y[index]).reshape(-1)[0])
        image_ms = self.MS[ms_x:ms_x + self.ms_size, ms_y:ms_y + self.ms_size, :].transpose((2, 0, 1))
        image_pan = self.PAN[s * ms_x:s * ms_x + self.pan_size, s * ms_y:s * ms_y + self.pan_size]
        image_pan = np.expand_dims(image_pan, axis=0) if image_pan.ndim == 2 else image_pan.transpose((2, 0, 1))
        label = torch.Tensor(self.Label[index]).squeeze()
        return (torch.from_numpy(np.ascontiguousarray(image_ms)).type(torch.FloatTensor),
                torch.from_numpy(np.ascontiguousarray(image_pan)).type(torch.FloatTensor), label, ms_x, ms_y)

    def __len__(self):
        return len(self.x)

    def index_view(self):
        return _IndexView(self)


class _IndexView(Dataset):
    def __init__(self, ds):
        self.x = np.asarray(ds.x).reshape(-1).astype(np.int64)
        self.y = np.asarray(ds.y).reshape(-1).astype(np.int64)
        self.label = np.asarray(ds.Label).reshape(-1).astype(np.int64)

    def __getitem__(self, index):
        return int(self.x[index]), int(self.y[index]), int(self.label[index]), int(index)

    def __getitems__(self, indices):
        """A whole batch at once, already collated (the loader twins pass it through: `collate_batched`): what the default
        collate makes of `[self[i] for i in indices]` — four int64 vectors — without a Python tuple per pixel (a 512x512 scene
        has 262,144 of them per epoch; the sampler, and with it the global RNG stream, is untouched)."""
        idx = np.asarray(indices, dtype=np.int64)
        return (torch.from_numpy(self.x[idx]), torch.from_numpy(self.y[idx]), torch.from_numpy(self.label[idx]), torch.from_numpy(idx))

    def __len__(self):
        return len(self.x)


def collate_batched(batch):
    """collate_fn of the index loaders: `_IndexView.__getitems__` has collated already."""
    return batch


class dataset_qua_dqtl(Dataset):
    """Four co-registered [H,W,4] images -> (ms, pan, ms_gan, pan_gan, label, x, y) (train/dataset.py:191-224)."""

    def __init__(self, ms, pan, ms_gan, pan_gan, xyl, cfg):
        self.images = (ms, pan, ms_gan, pan_gan)
        self.Label, self.x, self.y = xyl[2], xyl[0], xyl[1]
        self.size = cfg['patch_size']

    def index_view(self):
        return _IndexView(self)

    def __getitem__(self, index):
        p = self.size
        x, y = int(np.asarray(self.x[index]).reshape(-1)[0]), int(np.asarray(self.y[index]).reshape(-1)[0])
        out = [torch.from_numpy(np.ascontiguousarray(im[x:x + p, y:y + p, :].transpose((2, 0, 1)))).type(torch.FloatTensor)
               for im in self.images]
        label = torch.Tensor(self.Label[index]).squeeze()
        return out[0], out[1], out[2], out[3], label, x, y

    def __len__(self):
        return len(self.x)
