"""Oracle (test infrastructure): numpy restatement of the HIP path's counter-based dropout RNG.

The reference draws dropout masks from torch's Philox/MT streams (nn.Dropout, model/unet.py:21), which cannot be reproduced
on the device; the HIP path instead uses  h = fmix32((i>>1)*0x9E3779B1 + seed), r16 = (i&1) ? h>>16 : h&0xFFFF,
keep(i) = r16 >= floor(p*65536)  over the NHWC element index
(hpfg_amd/csrc/common.h).  This file restates that integer law bit-exactly so that (a) the device masks can be checked on
the CPU and (b) the oracle U-Net can be run with exactly the masks the kernels used.
"""
import numpy as np


def hash32(i: np.ndarray, seed: int) -> np.ndarray:
    h = (i.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def threshold(p: float) -> int:
    t = float(np.float32(p)) * 65536.0
    return 0xFFFF if t >= 65535.0 else int(t)


def keep_mask_nhwc(n_elems: int, p: float, seed: int) -> np.ndarray:
    i = np.arange(n_elems, dtype=np.uint64)
    h = hash32(i >> np.uint64(1), seed)
    r16 = np.where((i & np.uint64(1)) == 1, h >> np.uint32(16), h & np.uint32(0xFFFF))
    return (r16 >= np.uint32(threshold(p))).astype(np.uint8)


def keep_mask_nchw(n: int, c: int, h: int, w: int, p: float, seed: int) -> np.ndarray:
    """Mask in the oracle's NCHW layout for an activation whose device layout is NHWC."""
    return keep_mask_nhwc(n * h * w * c, p, seed).reshape(n, h, w, c).transpose(0, 3, 1, 2).copy()
