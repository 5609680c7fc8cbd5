"""Oracle (test infrastructure): CPU restatement of the hot-path losses.

  dice_loss      /root/reference/utils/loss/diceloss.py:155-191 (dup. medloss.py:5-41):
                 one-hot by equality, per class 1-(2*sum(p*t)+1e-5)/(sum(p*p)+sum(t*t)+1e-5)
                 with sums over the whole batch, mean over classes.
  med_sup_loss   /root/reference/utils/loss/medloss.py:44-56: ce*CE(ignore 255)+dice*Dice(softmax).
  mse_consistency  mean((p_s-p_t)^2), main.py:191 / 2017_03_NIPS_Mean-Teacher_ACDC.py:104.
  dense_loss     /root/reference/utils/loss/dense_loss.py:17-40 (NT-Xent, T=0.7).
  binary_dice    medpy ``metric.binary.dc`` (MedPy==0.4.0, requirements.txt:66; not vendored):
                 2*|A&B| / (|A|+|B|), 0.0 when both empty; call site val.py:376-387.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

SMOOTH = 1e-5


def dice_sums(prob: torch.Tensor, target: torch.Tensor):
    """prob [M,C,H,W]; target [M,1,H,W] or [M,H,W] (int or float class ids).
    Returns (I, Z, Y) each [C]: sum p*t, sum p*p, sum t*t over the batch."""
    if target.dim() == 3:
        target = target.unsqueeze(1)
    c = prob.shape[1]
    cls = torch.arange(c, dtype=torch.float32).view(1, c, 1, 1)
    onehot = (target.to(torch.float32) == cls).to(torch.float32)
    red = (0, 2, 3)
    return (prob * onehot).sum(red), (prob * prob).sum(red), (onehot * onehot).sum(red)


def dice_from_sums(i, z, y):
    return (1.0 - (2.0 * i + SMOOTH) / (z + y + SMOOTH)).mean()


def dice_loss(prob, target, softmax: bool = False):
    if softmax:
        prob = torch.softmax(prob, dim=1)
    return dice_from_sums(*dice_sums(prob, target))


def cross_entropy(logits, target, ignore_index: int = 255):
    return F.cross_entropy(logits, target.long(), ignore_index=ignore_index)


def med_sup_loss(logits, target, ce: float = 0.5, dice: float = 0.5):
    return ce * cross_entropy(logits, target) + dice * dice_loss(torch.softmax(logits, dim=1), target.unsqueeze(1))


def mse_consistency(prob_s, prob_t):
    return torch.mean((prob_s - prob_t) ** 2)


def nt_xent(a, b, temperature: float = 0.7):
    """a,b: [B,D] or [B,D,S]; L2-normalise over dim 1, flatten, SimCLR loss over the 2B rows."""
    a = F.normalize(a, dim=1).flatten(1)
    b = F.normalize(b, dim=1).flatten(1)
    n = a.shape[0]
    both = torch.cat([a, b], 0)
    sim = torch.exp(both @ both.t() / temperature)
    denom = sim.masked_fill(torch.eye(2 * n, dtype=torch.bool), 0.0).sum(-1)   # diagonal excluded, never subtracted
    pos = torch.exp((a * b).sum(-1) / temperature)
    pos = torch.cat([pos, pos], 0)
    return (-torch.log(pos / denom)).mean()


def dense_loss(x, y, temperature: float = 0.7):
    """x=(g,d) student, y=(g,d) teacher (detached): 0.5*(nt_xent(g)+nt_xent(d))."""
    return 0.5 * (nt_xent(x[0], y[0].detach(), temperature) + nt_xent(x[1], y[1].detach(), temperature))


def binary_dice(pred: np.ndarray, gt: np.ndarray) -> float:
    pred = np.asarray(pred).astype(bool)
    gt = np.asarray(gt).astype(bool)
    inter = np.count_nonzero(pred & gt)
    denom = np.count_nonzero(pred) + np.count_nonzero(gt)
    return 2.0 * inter / float(denom) if denom > 0 else 0.0


def mean_foreground_dice(pred_labels: np.ndarray, true_labels: np.ndarray, num_classes: int) -> float:
    """Per-class binary Dice for classes 1..C-1 with the reference's (0 if prediction empty)
    rule (val.py:376-387), averaged over classes."""
    vals = []
    for c in range(1, num_classes):
        p = pred_labels == c
        vals.append(binary_dice(p, true_labels == c) if p.sum() > 0 else 0.0)
    return float(np.mean(vals))
