"""Oracle (test infrastructure): host-side scalar laws of the hot loop, restated as pure functions.

  medical_lr(k)       lr in force for the k-th optimizer.step() (k = 1,2,...) under
                      /root/reference/utils/scheduler/medical_lr.py:13-17; the torch base class
                      calls step() once in its constructor, so last_epoch == k-1 at the k-th
                      optimizer step and lr = base*(1-(k-2)/max)^0.9 (first step slightly ABOVE base).
  cosine_table / cosine_lr(k)   /root/reference/utils/scheduler/warmup_cosine.py:19-38: table
                      lookup lr_schedule[last_epoch-1]; at construction last_epoch == 0 gives
                      index -1, i.e. the LAST table entry for the first optimizer step.
  sigmoid_rampup, linear_rampup, consistency weight   utils/utils.py:67-79,89-95.
  ema_alpha           utils/utils.py:84.
  box_masks           utils/utils.py:115-173 (CutMix box masks, numpy RNG draw order kept).
"""
from __future__ import annotations

import math

import numpy as np


def medical_lr(k: int, base_lr: float, max_iterations: int) -> float:
    return base_lr * (1.0 - (k - 2) / max_iterations) ** 0.9


def cosine_table(base_lr, warmup_epochs, warmup_lr, final_lr, iter_per_epoch, num_epochs):
    warm = np.linspace(warmup_lr, base_lr, iter_per_epoch * warmup_epochs)
    n = iter_per_epoch * (num_epochs - warmup_epochs) + 1
    cos = final_lr + 0.5 * (base_lr - final_lr) * (1 + np.cos(np.pi * np.arange(n) / n))
    return np.concatenate((warm, cos))


def cosine_lr(k: int, table: np.ndarray) -> float:
    """lr for the k-th optimizer step (k>=1): table[k-2] with python negative indexing."""
    return float(table[k - 2])


def sigmoid_rampup(current, length) -> float:
    if length == 0:
        return 1.0
    c = min(max(float(current), 0.0), float(length))
    ph = 1.0 - c / length
    return float(math.exp(-5.0 * ph * ph))


def linear_rampup(current, length) -> float:
    return 1.0 if current >= length else current / length


def ema_alpha(step: int, decay: float) -> float:
    return min(1.0 - 1.0 / (step + 1), decay)


def box_masks(n_masks, shape, rng, prop_range=(0.25, 0.5), n_boxes=4):
    """HPFG's configuration of the box-mask generator (main.py:94-108): area proportion,
    random aspect ratio, within bounds, inverted (start from zeros, XOR boxes)."""
    props = rng.uniform(prop_range[0], prop_range[1], size=(n_masks, n_boxes))
    zero = props == 0.0
    yp = np.exp(rng.uniform(0.0, 1.0, size=(n_masks, n_boxes)) * np.log(props))
    xp = props / yp
    fac = np.sqrt(1.0 / n_boxes)
    yp, xp = yp * fac, xp * fac
    yp[zero] = 0
    xp[zero] = 0
    sizes = np.round(np.stack([yp, xp], axis=2) * np.array(shape)[None, None, :])
    pos = np.round((np.array(shape) - sizes) * rng.uniform(0.0, 1.0, size=sizes.shape))
    rect = np.append(pos, pos + sizes, axis=2)
    m = np.zeros((n_masks, 1) + tuple(shape))
    for i in range(n_masks):
        for y0, x0, y1, x1 in rect[i]:
            sl = (i, 0, slice(int(y0), int(y1)), slice(int(x0), int(x1)))
            m[sl] = 1 - m[sl]
    return m
