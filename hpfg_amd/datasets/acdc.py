"""ACDC-format slice datasets from disk into the HBM-resident slice pool (reference datasets/ACDC.py:13-82,
get_acdc_loader :85-110, get_ssl_acdc_loader :115-135).

Directory contract of the reference (the SSL4MIS preprocessing): ``<root>/train_slices.list`` names 2-D training slices stored as
``<root>/data/slices/<case>.h5``, ``val.list`` / ``test.list`` name volumes ``<root>/data/<case>.h5``; every file holds ``image`` and ``label``.
The reference opens one h5 file per sample in DataLoader workers; here every file is read ONCE with the dependency-free reader
(``h5lite``, h5py is not part of the image), all training slices go into a ``DeviceSlicePool`` and the reference's RandomGenerator runs as
one gather kernel per batch (``device_pool``).  The labelled / unlabelled split is ``torch.utils.data.random_split``'s law: a permutation
from torch's default generator, first ``int(len * label_num)`` indices labelled (ACDC.py:124-127).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .h5lite import read_datasets


def _read_list(path: str) -> List[str]:
    with open(path, "r") as f:
        return [ln.replace("\n", "") for ln in f.readlines() if ln.strip()]


class ACDCFiles:
    """Host-side view of an ACDC root: file lists per split (ACDC.py:65-82) and arrays on demand."""

    def __init__(self, root: str, split: str = "train"):
        self.root, self.split = root, split
        if split == "train":
            self.sample_list = [f"{root}/data/slices/{c}.h5" for c in _read_list(root + "/train_slices.list")]
        elif split == "val":
            self.sample_list = [f"{root}/data/{c}.h5" for c in _read_list(root + "/val.list")]
        else:
            self.sample_list = [f"{root}/data/{c}.h5" for c in _read_list(root + "/test.list")]

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx) -> Tuple[np.ndarray, np.ndarray]:
        d = read_datasets(self.sample_list[idx], ("image", "label"))
        return np.array(d["image"], dtype=np.float32), np.array(d["label"], dtype=np.uint8)          # ACDC.py:40-41

    def label_to_img(self, label):          # datasets/ACDC.py:50-63
        from .synthetic import palette_image
        return palette_image(label, 4)


class _Volumes(torch.utils.data.Dataset):
    """bs=1 evaluation volumes (image [S,h,w] float32, label [S,h,w] uint8), read once."""

    def __init__(self, files: ACDCFiles):
        self.items = [files[i] for i in range(len(files))]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        img, lab = self.items[i]
        return torch.from_numpy(img), torch.from_numpy(lab)

    def label_to_img(self, label):          # datasets/ACDC.py:50-63
        from .synthetic import palette_image
        return palette_image(label, 4)


def _pool(root: str, device):
    from .device_pool import DeviceSlicePool
    files = ACDCFiles(root, "train")
    return DeviceSlicePool([files[i] for i in range(len(files))], device), len(files)


def get_acdc_loader(root: str, batch_size: int = 4, train_crop_size: Sequence[int] = (224, 224), device="cuda"):
    """(train_loader, test_loader) like the reference's get_acdc_loader (ACDC.py:85-110)."""
    from .device_pool import DevicePoolLoader
    pool, n = _pool(root, device)
    test = torch.utils.data.DataLoader(_Volumes(ACDCFiles(root, "test")), batch_size=1, shuffle=False)
    return DevicePoolLoader(pool, batch_size, train_crop_size), test


def get_ssl_acdc_loader(root: str, batch_size: int = 8, unlabel_batch_size: int = 24, train_crop_size: Sequence[int] = (224, 224), label_num: float = 0.2,
                        device="cuda"):
    """(label_loader, unlabel_loader, test_loader) like the reference's get_ssl_acdc_loader (ACDC.py:115-135)."""
    from .device_pool import DevicePoolLoader
    pool, n = _pool(root, device)
    label_length = int(n * label_num)
    perm = torch.randperm(n).tolist()                     # random_split: randperm(sum(lengths)) from the default generator
    lab_idx, unl_idx = perm[:label_length], perm[label_length:]
    test = torch.utils.data.DataLoader(_Volumes(ACDCFiles(root, "test")), batch_size=1, shuffle=False)
    return (DevicePoolLoader(pool, batch_size, train_crop_size, indices=lab_idx),
            DevicePoolLoader(pool, unlabel_batch_size, train_crop_size, indices=unl_idx), test)
