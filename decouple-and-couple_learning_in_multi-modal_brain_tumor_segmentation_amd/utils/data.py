"""N3 -- BraTS-shaped data path.  The reference's `data/` package (ClsWiseBraTS128.BraDataSet128,
train_no_amp.py:19,156-165) is absent from its repository; what its training loop consumes is fixed by the call site
(train_no_amp.py:184-189): per sample ``x [4, 128,128,128] float32, target [128,128,128] int64 in {0..3},
edge [128,128,128] int64 in {0,1,2,4,5,6,7,8} (tools.py:174-218), missing_modal``.

Two map-style datasets produce exactly that tuple from 240 x 240 x 155 volumes with a random 128^3 crop:
  * ``SyntheticBraTS`` -- generator-defined volumes (no files; utils.synthetic), used by bench / tests / the harness default;
  * ``NpzBraTS``       -- one ``.npz`` per subject with ``image [4,H,W,D]`` (or ``[H,W,D,4]``) float and ``label [H,W,D]`` integer
                          (BraTS labels 0,1,2,4 -- 4 is mapped to 3 as the reference's loaders do).  nibabel is not available in
                          this image, so NIfTI conversion is left to the user (one ``np.savez`` per subject).
Edge codes are derived from the label with utils.synthetic.edge_codes (boundary of each sub-region, coded per E1/E2/E4)."""
import glob
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import synthetic as syn

FULL_SIZE = (240, 240, 155)


def random_crop_origin(full, crop, rng):
    """Uniform crop origin; volumes smaller than the crop along an axis are zero-padded at the far end (D = 155 -> 160 is the
    reference's input_D, train_no_amp.py:63)."""
    return tuple(int(rng.integers(0, max(f - c, 0) + 1)) for f, c in zip(full, crop))


def crop_pad(vol, origin, crop):
    """vol [..., H, W, D] -> [..., crop]; zero padding where the crop leaves the volume."""
    out = vol.new_zeros(vol.shape[:-3] + tuple(crop))
    sl_src, sl_dst = [], []
    for o, c, f in zip(origin, crop, vol.shape[-3:]):
        n = max(min(c, f - o), 0)
        sl_src.append(slice(o, o + n)); sl_dst.append(slice(0, n))
    out[(Ellipsis,) + tuple(sl_dst)] = vol[(Ellipsis,) + tuple(sl_src)]
    return out


class SyntheticBraTS(Dataset):
    """`n_subjects` deterministic synthetic subjects; each access draws a fresh random crop (seeded by (seed, epoch, index))."""

    def __init__(self, n_subjects=8, crop=(128, 128, 128), seed=1000, full_size=None):
        self.n, self.crop, self.seed, self.epoch = int(n_subjects), tuple(crop), int(seed), 0
        self.full = tuple(full_size) if full_size is not None else tuple(crop)   # default: generate the patch directly

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if self.full == self.crop:
            x, target, edge = syn.synthetic_sample(i + 7919 * self.epoch, self.crop, self.seed)
            return x, target, edge, torch.zeros(4, dtype=torch.bool)
        x, target, _ = syn.synthetic_sample(i, self.full, self.seed)
        rng = np.random.default_rng([self.seed, self.epoch, i])
        o = random_crop_origin(self.full, self.crop, rng)
        x, target = crop_pad(x, o, self.crop), crop_pad(target, o, self.crop)
        return x, target, syn.edge_codes(target), torch.zeros(4, dtype=torch.bool)


class NpzBraTS(Dataset):
    def __init__(self, root, list_file=None, crop=(128, 128, 128), seed=1000, train=True):
        if list_file is not None:
            with open(list_file) as f:
                names = [ln.strip() for ln in f if ln.strip()]
            self.paths = [os.path.join(root, n if n.endswith(".npz") else n + ".npz") for n in names]
        else:
            self.paths = sorted(glob.glob(os.path.join(root, "*.npz")))
        if not self.paths:
            raise FileNotFoundError("no .npz subjects under %s" % root)
        self.crop, self.seed, self.epoch, self.train = tuple(crop), int(seed), 0, bool(train)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        with np.load(self.paths[i], allow_pickle=False) as z:
            img, lab = z["image"], z["label"]
        img = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
        if img.shape[-1] == 4 and img.shape[0] != 4:
            img = img.permute(3, 0, 1, 2).contiguous()
        lab = torch.from_numpy(np.ascontiguousarray(lab).astype(np.int64))
        lab[lab == 4] = 3
        if self.train:
            rng = np.random.default_rng([self.seed, self.epoch, i])
            o = random_crop_origin(tuple(lab.shape), self.crop, rng)
            img, lab = crop_pad(img, o, self.crop), crop_pad(lab, o, self.crop)
        return img, lab, syn.edge_codes(lab), torch.zeros(4, dtype=torch.bool)
