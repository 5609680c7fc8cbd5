"""Device-resident slice pool + training-time augmentation on the GPU (reference datasets/utils.py:73-117,
datasets/ACDC.py:36-48).

The reference reads one h5 slice per sample in DataLoader workers and augments it with numpy / scipy on the host
(``RandomGenerator``: with probability .5 rot90 + flip, else with probability .5 an integer-angle ``ndimage.rotate(order=0)``,
then ``zoom(order=0)`` to the network size).  An MI355X holds every ACDC slice in a sliver of its 288 GB, so here the pool lives
in HBM and a batch is ONE gather kernel (``hpfg_augment_batch``): the host only draws the random parameters -- in the
reference's order, from the same ``random`` / ``numpy.random`` generators -- and ships a few dozen bytes per sample.
Results are bit-identical to the host pipeline (tests/test_gpu_augment.py checks against oracle/augment_ref.py).
"""
from __future__ import annotations

import ctypes as C
import random as _pyrandom
from functools import lru_cache
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L


@lru_cache(maxsize=256)
def _zoom_table(n_in: int, n_out: int) -> np.ndarray:
    """Source index of every output index along one axis for scipy.ndimage.zoom(a, n_out / n_in, order=0) (separable); -1 where
    scipy writes its constant 0 instead (its last coordinate can land a rounding error beyond n_in - 1, mode='constant')."""
    from scipy.ndimage import zoom
    t = zoom(np.arange(1, n_in + 1, dtype=np.float64), n_out / n_in, order=0)
    assert t.shape == (n_out,), (t.shape, n_in, n_out)
    return np.rint(t).astype(np.int32) - 1


def _rotate_params(h: int, w: int, angle: int) -> Tuple[float, float, float, float, float, float]:
    """Matrix and offset of scipy.ndimage.rotate(a, angle, reshape=False) for a 2-D array, computed the way scipy does."""
    from scipy import special
    c, s = special.cosdg(angle), special.sindg(angle)
    m = np.array([[c, s], [-s, c]])
    shp = np.array([h, w])
    out_center = m @ ((shp - 1) / 2)
    in_center = (shp - 1) / 2
    off = in_center - out_center
    return float(m[0, 0]), float(m[0, 1]), float(m[1, 0]), float(m[1, 1]), float(off[0]), float(off[1])


class DeviceSlicePool:
    """All (image [h,w] float32, mask [h,w] uint8) slices of a dataset, concatenated in two device buffers."""

    def __init__(self, slices: Sequence[Tuple[np.ndarray, np.ndarray]], device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceSlicePool lives in GPU memory (no CPU fallback)")
        self.shapes: List[Tuple[int, int]] = []
        self.offsets: List[int] = []
        imgs, labs, off = [], [], 0
        for img, lab in slices:
            img = np.ascontiguousarray(img, dtype=np.float32)
            lab = np.ascontiguousarray(lab, dtype=np.uint8)
            assert img.ndim == 2 and img.shape == lab.shape
            self.shapes.append(img.shape)
            self.offsets.append(off)
            off += img.size
            imgs.append(img.reshape(-1))
            labs.append(lab.reshape(-1))
        self.images = torch.from_numpy(np.concatenate(imgs)).to(self.device)
        self.labels = torch.from_numpy(np.concatenate(labs)).to(self.device)

    def __len__(self):
        return len(self.shapes)


class RandomGeneratorDevice:
    """``RandomGenerator(output_size)`` of the reference (datasets/utils.py:99-117), evaluated on the device for a whole batch."""

    def __init__(self, output_size: Sequence[int]):
        self.output_size = (int(output_size[0]), int(output_size[1]))

    def draw(self, h: int, w: int, py_rng=_pyrandom, np_rng=np.random):
        """One sample's random decisions, in the reference's order: (mode, k, axis, angle)."""
        if py_rng.random() > 0.5:
            k = int(np_rng.randint(0, 4))
            axis = int(np_rng.randint(0, 2))
            return 1, k, axis, 0
        if py_rng.random() > 0.5:
            return 2, 0, 0, int(np_rng.randint(-20, 20))
        return 0, 0, 0, 0

    def __call__(self, pool: DeviceSlicePool, indices: Sequence[int], py_rng=_pyrandom, np_rng=np.random, stream: Optional[int] = None):
        H, W = self.output_size
        B = len(indices)
        samples = (L.AugSample * B)()
        tabs = np.empty(B * (H + W), dtype=np.int32)
        for b, idx in enumerate(indices):
            h, w = pool.shapes[idx]
            mode, k, axis, angle = self.draw(h, w, py_rng, np_rng)
            s = samples[b]
            s.img_off = s.lab_off = pool.offsets[idx]
            s.h, s.w, s.mode, s.k, s.axis = h, w, mode, k, axis
            h1, w1 = (w, h) if (mode == 1 and k % 2 == 1) else (h, w)      # rot90 by an odd k swaps the axes before the zoom
            s.tab_off = b * (H + W)
            tabs[s.tab_off:s.tab_off + H] = _zoom_table(h1, H)
            tabs[s.tab_off + H:s.tab_off + H + W] = _zoom_table(w1, W)
            if mode == 2:
                s.m00, s.m01, s.m10, s.m11, s.off_y, s.off_x = _rotate_params(h, w, angle)
        dev = pool.device
        s_dev = torch.frombuffer(bytearray(bytes(samples)), dtype=torch.uint8).to(dev, non_blocking=True)
        t_dev = torch.from_numpy(tabs).to(dev, non_blocking=True)
        image = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        mask = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        L.check(L.load().hpfg_augment_batch(L.ptr(pool.images), L.ptr(pool.labels), s_dev.data_ptr(), L.ptr(t_dev), B, H, W, L.ptr(image),
                                            L.ptr(mask), st), "augment_batch")
        return image, mask


class DevicePoolLoader:
    """DataLoader-shaped iterator over a DeviceSlicePool: ``shuffle=True, drop_last=True`` batches of
    ``(image float32 [B,1,H,W], mask uint8 [B,H,W])`` (the reference's train-batch contract, datasets/ACDC.py:127-129), produced on
    the device by RandomGeneratorDevice; ``len(loader)``, ``len(loader.dataset)`` and StopIteration restart behave like the
    reference's loaders (main.py:127-135)."""

    def __init__(self, pool: DeviceSlicePool, batch_size: int, output_size: Sequence[int], shuffle: bool = True, indices: Optional[Sequence[int]] = None):
        self.dataset, self.batch_size, self.shuffle = pool, int(batch_size), shuffle
        self.indices = list(range(len(pool))) if indices is None else list(indices)
        self.gen = RandomGeneratorDevice(output_size)

    def __len__(self):
        return len(self.indices) // self.batch_size

    def __iter__(self):
        order = list(np.random.permutation(self.indices)) if self.shuffle else list(self.indices)
        for i in range(len(self)):
            yield self.gen(self.dataset, [int(j) for j in order[i * self.batch_size:(i + 1) * self.batch_size]])
