"""Learning-rate laws of the hot loop, reproducing the reference's quirks.

``Medical_LR`` (utils/scheduler/medical_lr.py:13-17): lr = base*(1-(last_epoch-1)/max)^0.9; torch's scheduler base class
steps once at construction, so the first optimizer step runs slightly ABOVE base_lr.
``CosineWarmupLR_Scheduler`` (utils/scheduler/warmup_cosine.py:19-38): table lookup lr_schedule[last_epoch-1]; at
construction last_epoch == 0 indexes the LAST entry, so the first optimizer step runs at ~final_lr.
``PolyLR`` (utils/scheduler/poly.py:8-14).
They are host scalars only: they write ``param_groups[i]['lr']``; FusedSGD copies that to the device asynchronously.
"""
from __future__ import annotations

import numpy as np
from torch.optim.lr_scheduler import _LRScheduler


class Medical_LR(_LRScheduler):
    def __init__(self, optimizer, base_lr, max_iterations):
        self.base_lr, self.max_iterations = base_lr, max_iterations
        super().__init__(optimizer, last_epoch=-1)

    def get_lr(self):
        lr = self.base_lr * (1.0 - (self.last_epoch - 1) / self.max_iterations) ** 0.9
        return [lr] * len(self.base_lrs)


class CosineWarmupLR_Scheduler(_LRScheduler):
    def __init__(self, optimizer, warmup_epochs=10, warmup_lr=1e-6, num_epochs=100, base_lr=0.01, final_lr=1e-6, iter_per_epoch=1000):
        self.base_lr = base_lr
        n_warm = iter_per_epoch * warmup_epochs
        n_decay = iter_per_epoch * (num_epochs - warmup_epochs) + 1
        warm = np.linspace(warmup_lr, base_lr, n_warm)
        decay = final_lr + 0.5 * (base_lr - final_lr) * (1 + np.cos(np.pi * np.arange(n_decay) / n_decay))
        self.lr_schedule = np.concatenate((warm, decay))
        super().__init__(optimizer, last_epoch=-1)

    def get_lr(self):
        lr = self.lr_schedule[self.last_epoch - 1]      # index -1 on the very first call: reference behaviour
        return [lr] * len(self.base_lrs)


class PolyLR(_LRScheduler):
    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power, self.max_iters, self.min_lr = power, max_iters, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [max(b * (1 - self.last_epoch / self.max_iters) ** self.power, self.min_lr) for b in self.base_lrs]
