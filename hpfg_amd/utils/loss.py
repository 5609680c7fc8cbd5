"""Segmentation losses of the hot path on the HIP library.

Mirrors the reference's loss API for this path: ``DiceLoss(n_classes)(inputs, target, weight=None, softmax=False)``
(utils/loss/diceloss.py:155-191), ``Med_Sup_Loss(num_classes, ce=.5, dice=.5)(outputs, target)`` (utils/loss/medloss.py:44-56),
``Dense_Loss(batch_size, device, temperature=.7)(x, y)`` (utils/loss/dense_loss.py:5-40), plus the fused step loss
``seg_loss`` that the step drivers use (softmax + CE + Dice + MSE consistency in two kernels, no host synchronisation).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib as L


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] logical tensor -> contiguous [N,H,W,C] storage (zero-copy when it already is channels-last)."""
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _labels_u8(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dim() == 4:
        t = t[:, 0]
    if t.dtype != torch.uint8:
        t = t.to(torch.uint8)       # class ids and the ignore value 255 fit; float pseudo-labels hold integers (main.py:178)
    return t.contiguous()


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, coef, labels0, labels1, t_logits, n_lab, is_prob, dp, t_is_prob=False, cons_mask=None):
        lib = L.load()
        if not logits.is_cuda:
            raise RuntimeError("hpfg_amd losses run on the HIP library only (no CPU fallback)")
        x = _nhwc(logits.float())
        N, H, W, Cc = x.shape
        dev = x.device
        a = L.LossArgs()
        nblk = lib.hpfg_loss_blocks(N, H, W)
        partials = torch.empty(nblk * L.LOSS_NSUM, dtype=torch.float32, device=dev)
        sums = torch.empty(L.LOSS_NSUM, dtype=torch.float32, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        t = _nhwc(t_logits.float()) if t_logits is not None else None
        a.logits, a.t_logits, a.labels0, a.labels1 = L.ptr(x), L.ptr(t), L.ptr(labels0), L.ptr(labels1)
        a.coef, a.partials, a.sums, a.out, a.dlogits = L.ptr(coef), L.ptr(partials), L.ptr(sums), L.ptr(out), None
        a.N, a.n_lab, a.H, a.W, a.C = N, n_lab, H, W, Cc
        if dp is not None and not getattr(dp, "sync_bn", True):
            dp = None                     # per-rank loss (DDP semantics): no exchange of the partial sums
        a.world = dp.world_size if dp is not None else 1
        a.input_is_prob = 1 if is_prob else 0
        a.teacher_is_prob = 1 if t_is_prob else 0
        if t is not None:
            if t.shape[0] == N - n_lab and n_lab > 0:
                a.t_unlab_only = 1
            elif t.shape[0] != N:
                raise ValueError(f"consistency target has {t.shape[0]} images; expected {N} (whole batch) or {N - n_lab} (unlabelled part)")
        cm = None
        if cons_mask is not None:
            cm = cons_mask.float().reshape(-1).contiguous()
            if t is None or cm.numel() != (N - n_lab) * H * W:
                raise ValueError("cons_mask needs a consistency target and one weight per unlabelled pixel")
            a.cons_mask = L.ptr(cm)
        st = torch.cuda.current_stream(dev).cuda_stream
        if dp is not None and getattr(dp, "p2p", False) and (dp.world_size > 1 or dp.force_sync):
            slot, epoch = dp.next_loss_slot()      # (a slot and an epoch word per loss call of the step: no two calls share one)
            dp.bump(epoch, st)          # the ranks' loss sums are added by the reduction kernel itself (peer mailboxes)
            px = dp.peer_desc(slot, epoch)
            L.check(lib.hpfg_seg_loss_partials_x(C.byref(a), C.byref(px), st), "seg_loss_partials_x")
        else:
            L.check(lib.hpfg_seg_loss_partials(C.byref(a), st), "seg_loss_partials")
            if dp is not None and (dp.world_size > 1 or dp.force_sync):
                dp.allreduce_sum(sums)
        L.check(lib.hpfg_seg_loss_finalize(C.byref(a), st), "seg_loss_finalize")
        ctx.args, ctx.keep = a, (x, t, labels0, labels1, coef, sums, cm)
        ctx.shape = (N, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        a = ctx.args
        N, H, W, Cc = ctx.shape
        x = ctx.keep[0]
        dl = torch.empty(N, H, W, Cc, dtype=torch.float32, device=x.device)
        a.dlogits = L.ptr(dl)
        gs = gout[0:1].contiguous()     # d(total)/d(out[0]); the other entries are detached diagnostics
        L.check(lib.hpfg_seg_loss_bwd(C.byref(a), L.ptr(gs), torch.cuda.current_stream(x.device).cuda_stream), "seg_loss_bwd")
        return dl.permute(0, 3, 1, 2), None, None, None, None, None, None, None, None, None


def seg_loss(logits: torch.Tensor, labels: Optional[torch.Tensor], n_lab: Optional[int] = None, *,
             coef: torch.Tensor, pseudo: Optional[torch.Tensor] = None, teacher_logits: Optional[torch.Tensor] = None,
             is_prob: bool = False, dp=None, teacher_prob: Optional[torch.Tensor] = None,
             cons_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused loss over logits [N,C,H,W].

    coef (device fp32 [8]) = [ce0, dice0, ce1, dice1, mse_w, 0, 0, 0]; images [0,n_lab) use ``labels`` (group 0),
    images [n_lab,N) use ``pseudo`` (group 1) and, if ``teacher_logits`` is given, the MSE between the two softmaxes.
    ``teacher_prob`` (probabilities, e.g. ICT's mixed teacher prediction) replaces ``teacher_logits`` as the MSE target.  Either
    target covers the whole batch [N,...] or just the unlabelled images [N-n_lab,...].  ``cons_mask`` ([N-n_lab,1,H,W] 0/1) turns
    the mean into UAMT's masked form  sum(mask*d^2) / (2*sum(mask) + 1e-16)  (2019_07_MICCAI_Uncertainty_Aware_ACDC.py:160-164).
    Returns a device tensor [8] = [total, ce0, dice0, ce1, dice1, mse, 0, 0]; only [0] carries gradient.
    """
    N = logits.shape[0]
    n_lab = N if n_lab is None else int(n_lab)
    if teacher_prob is not None:
        assert teacher_logits is None
        return _SegLossFn.apply(logits, coef, _labels_u8(labels), _labels_u8(pseudo), teacher_prob, n_lab, is_prob, dp, True, cons_mask)
    return _SegLossFn.apply(logits, coef, _labels_u8(labels), _labels_u8(pseudo), teacher_logits, n_lab, is_prob, dp, False, cons_mask)


def _coef(dev, vals: Sequence[float]) -> torch.Tensor:
    v = list(vals) + [0.0] * (8 - len(vals))
    return torch.tensor(v, dtype=torch.float32, device=dev)


class DiceLoss(nn.Module):
    """Batch-wide soft Dice, reference signature (diceloss.py:178): inputs are probabilities unless softmax=True."""

    def __init__(self, n_classes: int):
        super().__init__()
        self.n_classes = n_classes
        self._coef = {}

    def forward(self, inputs, target, weight=None, softmax=False):
        if weight is not None and any(w != 1 for w in weight):
            raise NotImplementedError("class weights are not used on the hot path")
        assert inputs.shape[1] == self.n_classes, "predict & target shape do not match"
        k = inputs.device
        if k not in self._coef:
            self._coef[k] = _coef(k, [0.0, 1.0])
        return seg_loss(inputs, target, coef=self._coef[k], is_prob=not softmax)[0]


class Med_Sup_Loss(nn.Module):
    def __init__(self, num_classes: int, ce: float = 0.5, dice: float = 0.5):
        super().__init__()
        self.num_classes, self.ce, self.dice = num_classes, ce, dice
        self._coef = {}

    def forward(self, outputs, target_label):
        k = outputs.device
        if k not in self._coef:
            self._coef[k] = _coef(k, [self.ce, self.dice])
        return seg_loss(outputs, target_label, coef=self._coef[k])[0]


class Dense_Loss(nn.Module):
    """NT-Xent between student and (detached) teacher neck features (dense_loss.py:17-40) on the HIP library: L2 normalisation over
    dim 1, the [2N,2N] Gram matrix (hpfg_gemm_f32), the row losses and the gradient of the student features (hpfg_amd.heads.ntxent)."""

    def __init__(self, batch_size: int = 32, device=None, temperature: float = 0.7):
        super().__init__()
        self.batch_size, self.temperature = batch_size, temperature
        self.dp = None      # data parallel, global-batch mode: the step sets its DataParallelContext and the contrast spans all ranks' features

    def contrastive_loss(self, a, b):
        from ..heads import ntxent
        return ntxent(a, b, self.temperature, self.dp)

    def forward(self, x, y):
        return 0.5 * (self.contrastive_loss(x[0], y[0].detach()) + self.contrastive_loss(x[1], y[1].detach()))
