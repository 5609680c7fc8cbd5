"""Deterministic synthetic stand-in for the reference's slice datasets (datasets/ACDC.py:36-48, datasets/utils.py:99-117).

Keeps the batch contract the training loops rely on: images float32 [B,C,H,W], masks uint8 [B,H,W], ``drop_last`` /
``shuffle`` loaders, ``len(loader)``, ``len(loader.dataset)``, ``iter()/next()`` with StopIteration restart (main.py:127-135).
Samples follow SURVEY.md section 8(d): a coarse random label map (cell x cell blocks) so classes are spatially coherent and Dice is
non-degenerate, image = label/(ncls-1) + 0.1*noise, broadcast over the input channels.
"""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader, Dataset



def palette_image(label, ncls: int = 4):
    """``label_to_img`` of the reference's datasets (datasets/ACDC.py:50-63) for one label map: class index -> colour, uint8 [H,W,3]
    (ignore label 255 shown as background).  Classes 0 .. 3 take the reference's ACDC colours; further classes a fixed spread of hues."""
    import numpy as np
    import torch
    if isinstance(label, torch.Tensor):
        label = label.cpu().numpy()
    lab = np.array(label).astype(np.uint8)
    lab[lab == 255] = 0
    pal = np.zeros((max(ncls, int(lab.max()) + 1, 4), 3), dtype=np.uint8)
    pal[:4] = [[0, 0, 0], [0, 0, 255], [0, 255, 0], [255, 0, 0]]
    for c in range(4, pal.shape[0]):
        pal[c] = [(53 * c) % 256, (97 * c) % 256, (193 * c) % 256]
    return pal[lab]

def synth_batch(seed: int, n: int, h: int, w: int, in_ch: int = 1, ncls: int = 4, cell: int = 32):
    """(images float32 [n,in_ch,h,w], labels uint8 [n,h,w]) from a CPU generator seeded with `seed`."""
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, ncls, (n, h // cell, w // cell), generator=g)
    lab = lab.repeat_interleave(cell, 1).repeat_interleave(cell, 2)
    img = lab.to(torch.float32).unsqueeze(1) / max(ncls - 1, 1) + 0.1 * torch.randn(n, 1, h, w, generator=g)
    return img.expand(n, in_ch, h, w).contiguous(), lab.to(torch.uint8)


class SyntheticSlices(Dataset):
    def __init__(self, n: int, size, in_ch: int = 1, ncls: int = 4, seed: int = 1234, cell: int = None):
        h, w = (size, size) if isinstance(size, int) else (size[0], size[1])
        cell = cell or max(h // 7, 1)
        while h % cell or w % cell:
            cell -= 1
        self.images, self.labels = synth_batch(seed, n, h, w, in_ch, ncls, cell)
        self.num_classes = ncls

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        return self.images[i], self.labels[i]

    def label_to_img(self, label):   # used only for TensorBoard images in the reference (main.py:318)
        return palette_image(label, self.num_classes)


class SyntheticVolumes(Dataset):
    """bs=1 test volumes (image [S,h,w], label [S,h,w]) like the reference's test split (val.py:154-193)."""

    def __init__(self, n_vol: int, slices: int, size, ncls: int = 4, seed: int = 4321):
        h, w = (size, size) if isinstance(size, int) else (size[0], size[1])
        cell = max(h // 7, 1)
        while h % cell or w % cell:
            cell -= 1
        img, lab = synth_batch(seed, n_vol * slices, h, w, 1, ncls, cell)
        self.images = img[:, 0].view(n_vol, slices, h, w)
        self.labels = lab.view(n_vol, slices, h, w)

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        return self.images[i], self.labels[i]

    def label_to_img(self, label):
        return palette_image(label)


def _size(args):
    s = args.train_crop_size
    return (s, s) if isinstance(s, int) else (s[0], s[1])


def get_ssl_synthetic_loader(args, rank: int = 0):
    n_lab = int(args.get("synthetic_labeled", 256))
    n_unl = int(args.get("synthetic_unlabeled", 1024))
    in_ch = int(args.get("in_channels", args.get("model1", {}).get("in_channels", 1)) if "in_channels" in args or "model1" in args else 1)
    if "model1" in args and "in_channels" in args.model1:
        in_ch = int(args.model1.in_channels)
    ncls = int(args.num_classes)
    seed = 1234 + rank
    lab = SyntheticSlices(n_lab, _size(args), in_ch, ncls, seed)
    unl = SyntheticSlices(n_unl, _size(args), in_ch, ncls, seed + 100003)
    test = SyntheticVolumes(int(args.get("synthetic_test_volumes", 2)), 8, _size(args), ncls, seed + 200003)
    g = torch.Generator().manual_seed(seed)
    label_loader = DataLoader(lab, batch_size=args.batch_size, shuffle=True, drop_last=True, num_workers=0, generator=g)
    unlabel_loader = DataLoader(unl, batch_size=args.unlabel_batch_size, shuffle=True, drop_last=True, num_workers=0, generator=g)
    test_loader = DataLoader(test, batch_size=1, shuffle=False)
    return label_loader, unlabel_loader, test_loader


def get_synthetic_loader(args, rank: int = 0):
    n = int(args.get("synthetic_labeled", 8))
    ncls = int(args.num_classes)
    seed = 1234 + rank
    ds = SyntheticSlices(n, _size(args), int(args.get("in_channels", 1)), ncls, seed)
    test = SyntheticVolumes(int(args.get("synthetic_test_volumes", 2)), 8, _size(args), ncls, seed + 200003)
    g = torch.Generator().manual_seed(seed)
    return (DataLoader(ds, batch_size=args.batch_size, shuffle=True, drop_last=True, num_workers=0, generator=g),
            DataLoader(test, batch_size=1, shuffle=False))
