"""Per-iteration step laws of the reference's drivers, composed from the HIP path.

  SupervisedStep    sup_ACDC.py:83-93            (Supervise)
  MeanTeacherStep   2017_03_NIPS_Mean-Teacher_ACDC.py:82-113   (Mean_Teacher)
  CPSStep           2021_06_CVPR_CPS_ACDC.py:83-120            (CPS)
  HPFGStep          main.py:125-212              (HPFG, incl. update_ema_variables_backbone main.py:68-76)
  ICTStep / UAMTStep / CTCTStep / S4CVNetStep   the loop bodies of 2022_02_ISBI_ICT-MedSeg_ACDC.py, 2019_07_MICCAI_Uncertainty_Aware_ACDC.py,
                    2021_12_MIDL_CTCT_ACDC.py (CTCT) and 2022_08_CVPR_S4CVNet_ACDC.py on the same kernels

Each step object owns the optimizer(s) / scheduler(s) built by the reference-compatible factories and exposes
``step(batch..., cur_itrs) -> dict of device scalars``.  Nothing in a step synchronises with the host: losses stay on the
device (the reference's ``loss.item()`` / per-class ``dice.item()`` calls have no counterpart); per-step scalars (learning rates,
consistency weight, EMA alpha) are staged through one pinned buffer and read by the kernels from device memory, so a whole
step can be captured into a hipGraph (``GraphedStep``).

The driver loops at the bottom keep the reference's function names and signatures.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from .utils import (BoxMaskGenerator, build_lr_scheduler, build_optimizer, ema_alpha, linear_rampup, seg_loss, sigmoid_rampup,
                    update_ema_variables, update_ema_variables_backbone)
from .utils.loss import _nhwc
from .utils.optim import FusedSGD

# layout of the per-step scalar block (device fp32 [32])
S_LR1, S_LR2, S_ALPHA, S_THRESH, S_COEF_A, S_COEF_B = 0, 1, 2, 3, 8, 16


class StepScalars:
    """Host-computed per-iteration scalars -> device, one async copy per step.

    ``host`` is a plain staging block the step laws fill in; ``push()`` snapshots it into the next slot of a ring of pinned
    blocks and copies THAT slot to the device.  A slot is reused only after the event recorded behind its copy has completed, so
    a host that runs several steps ahead of the GPU (hipGraph replays take ~50 us of host time against ms of GPU time) can never
    overwrite scalars a queued copy has not read yet.  The copy is issued eagerly in front of a graph replay (never captured: a
    captured memcpy node would re-read one fixed host address at execution time)."""

    SLOTS = 8

    def __init__(self, dev):
        self.host = torch.zeros(32, dtype=torch.float32)
        self.ring = torch.zeros(self.SLOTS, 32, dtype=torch.float32).pin_memory()
        self.events = [None] * self.SLOTS
        self.slot = 0
        self.dev = torch.zeros(32, dtype=torch.float32, device=dev)
        self.on_push = None          # called at the beginning of every step (data parallel: the per-step loss-slot counter restarts)

    def push(self):
        if self.on_push is not None:
            self.on_push()
        k = self.slot
        self.slot = (k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copy that last read this slot has executed
        self.ring[k].copy_(self.host)
        self.dev.copy_(self.ring[k], non_blocking=True)
        ev = self.events[k] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev.device))
        self.events[k] = ev

    def view(self, off, n=1):
        return self.dev[off:off + n]


def cat_batch(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], 0) -- without the copy when the two batches already sit back to back in one buffer (loaders and bench.py hand the
    labelled and unlabelled images over as adjacent views): the result is then a view over both."""
    if (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:] and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.storage_offset() == a.storage_offset() + a.numel()):
        return a.as_strided((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat([a, b], 0)


def batch_pair(a: torch.Tensor, b: torch.Tensor):
    """Copies of (a, b) laid out back to back in one buffer, so that cat_batch(a', b') is free (call once, outside the step loop)."""
    buf = torch.cat([a, b], 0)
    return buf[:a.shape[0]], buf[a.shape[0]:]


def argmax_labels(logits: torch.Tensor, mix_labels: Optional[torch.Tensor] = None, mix_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [N,H,W] arg-max over classes of logits [N,C,H,W]; with (mix_labels, mix_mask): labels*(1-M) + argmax*M."""
    x = _nhwc(logits.detach())
    N, H, W, Cc = x.shape
    out = torch.empty(N, H, W, dtype=torch.uint8, device=x.device)
    ml = mix_labels.contiguous() if mix_labels is not None else None
    mm = mix_mask.contiguous().float() if mix_mask is not None else None
    L.check(L.load().hpfg_argmax_labels(L.ptr(x), N, H, W, Cc, L.ptr(ml), L.ptr(mm), L.ptr(out), torch.cuda.current_stream(x.device).cuda_stream),
            "argmax_labels")
    return out


def cutmix_blend(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """a*(1-M) + b*M elementwise (main.py:149); mask broadcast over channels is materialised by the caller if C > 1."""
    a, b = a.contiguous(), b.contiguous()
    m = mask.expand_as(a).contiguous()
    out = torch.empty_like(a)
    L.check(L.load().hpfg_cutmix_blend(L.ptr(a), L.ptr(b), L.ptr(m), L.ptr(out), a.numel(), torch.cuda.current_stream(a.device).cuda_stream),
            "cutmix_blend")
    return out


class _StepBase:
    def __init__(self, dev, dp=None):
        self.dev, self.dp = dev, dp
        self.sc = StepScalars(dev)
        if dp is not None:
            self.sc.on_push = lambda: setattr(dp, "_loss_call", 0)
        # HPFG_STEP_MARKS=1: timestamp kernels at phase boundaries of the step (tools/stream_timeline.py); off by default
        self.marks = torch.zeros(32, dtype=torch.int64, device=dev) if os.environ.get("HPFG_STEP_MARKS", "0") == "1" else None
        # the no-grad teacher forward is independent of the student forward: run it on a second HIP stream so that the two
        # kernel chains (and, data parallel, their small BatchNorm collectives) overlap; joins before the loss.
        self.overlap = os.environ.get("HPFG_OVERLAP", "1") == "1"
        self.side = torch.cuda.Stream(device=dev) if self.overlap else None

    def _mark(self, i):
        if self.marks is not None:
            L.check(L.load().hpfg_timestamp(self.marks.data_ptr() + 8 * i, torch.cuda.current_stream(self.dev).cuda_stream), "timestamp")

    def _teacher_forward(self, ema_model, x):
        """ema_model(x) under no_grad, on the side stream when overlap is enabled.  Returns the teacher outputs."""
        if not self.overlap:
            with torch.no_grad():
                return ema_model(x)
        cur = torch.cuda.current_stream(self.dev)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side), torch.no_grad():
            self._mark(2)
            out = ema_model(x)
            self._mark(3)
        self._pending_join = True
        return out

    def _join_teacher(self, *outs):
        if self.overlap and getattr(self, "_pending_join", False):
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.side)
            for o in outs:
                for t_ in (o if isinstance(o, (tuple, list)) else (o,)):
                    for u in (t_ if isinstance(t_, (tuple, list)) else (t_,)):
                        if torch.is_tensor(u):
                            u.record_stream(cur)
            self._pending_join = False

    def _join_backward(self, stream):
        """A network whose forward ran on `stream` back-propagates there too (autograd runs a node on the stream of its forward).  Its
        backward node hands autograd no gradient tensors (they are written in place into the flat buffer), so autograd has nothing to
        synchronise that stream with: the step joins it explicitly before anything reads the gradients -- and, inside a capture, before
        the capture ends (an unjoined forked stream is what made hipStreamEndCapture of the HPFG step fault)."""
        if stream is not None and self.overlap:
            torch.cuda.current_stream(self.dev).wait_stream(stream)

    def _attach(self, model, alone: bool = False):
        """alone: the only trainable network of the step -- its backward may fork its off-chain weight gradients onto a side stream
        (UNetEngine.backward).  Not with two networks training on two streams: the second network's backward runs on a FORKED stream, and
        an event wait between two non-origin streams of a capture makes hipStreamEndCapture fault on this stack (ROCm 7.2; 15-line
        reproduction without any of this repository's code: tools/nested_fork_probe.py -- origin -> A, origin -> B, B waits for A, both
        joined: segmentation fault in capture_end; every fork / join against the origin stream itself is fine).  So inside a capture every
        side stream forks from and joins into the stream the capture began on, and nothing else."""
        model.dp = self.dp
        model._alone = bool(alone)          # (UNet's backward node: the peer-window gradient buckets fork a side stream only from the step's origin stream)
        if alone:
            model.defer_wgrad = True
        if any(p.requires_grad for p in model.parameters()):
            model.direct_grads = True          # one zero_grad + one backward per step: write gradients in place (no memset, no add)

    def _sgd_ema_update(self):
        """optimizer.step(); update_ema_variables(model, ema_model, ...) -- the tail of every Mean-Teacher-family iteration
        (2017_03_NIPS_Mean-Teacher_ACDC.py:108-113) -- as ONE launch over the student's flat buffers where the optimizer offers it."""
        fs, ft = getattr(self.model, "flat_params", None), getattr(self.ema_model, "flat_params", None)
        if isinstance(self.optimizer, FusedSGD) and fs is not None and ft is not None and fs.numel() == ft.numel() and fs.is_cuda:
            self.optimizer.step(push_lr=False, ema=(ft, fs.numel(), self.sc.view(S_ALPHA)))
        else:
            self.optimizer.step(push_lr=False)
            update_ema_variables(self.model, self.ema_model, self.args.ema_decay, 0, alpha_dev=self.sc.view(S_ALPHA))

    def _loss_backward(self, res):
        """backward() of the fused loss vector [total, parts...]: only element 0 carries gradient; a constant one-hot gradient
        spares autograd a ones() + zeros() + select-scatter per step."""
        if getattr(self, "_g1", None) is None or self._g1.device != res.device:
            self._g1 = torch.zeros(res.numel(), dtype=res.dtype, device=res.device)
            self._g1[0] = 1.0
        res.backward(self._g1)

    def _reduce_grads(self, *models):
        if self.dp is not None and (self.dp.world_size > 1 or self.dp.force_sync):
            p2p = getattr(self.dp, "p2p_grads", False)      # the exchange as kernels on this stream (peer windows): capturable, no RCCL call
            red = self.dp.peer_allreduce_sum if p2p else self.dp.allreduce_sum
            for m in models:
                if getattr(m, "_buckets_launched", False):      # the engine's backward already handed both buckets to the side stream
                    m._buckets_launched = False
                    self.dp.join_buckets()
                elif hasattr(m, "flat_grads") and not hasattr(m, "_hpfg_generic_flat"):
                    red(m.flat_grads)
                elif getattr(m, "_hpfg_flat_optimizer", None) is not None:
                    # SegFormer: autograd leaves per-parameter gradients; FusedAdamW packs them into the model's flat gradient buffer with ONE
                    # concat (what its step() does anyway) -- the exchange reduces that buffer in place and the step consumes it: no copy back
                    opt = m._hpfg_flat_optimizer      # (external_gather was set when the step was built: its step() consumes flat_grads as reduced here)
                    assert opt.external_gather, "a flat-buffer optimizer under data parallel must be registered with _StepBase._own_gather"
                    red(opt.gather_flat_grads())
                else:                      # any other module: one flattened exchange of the SUM, copied back (the 1/world of the per-rank-BatchNorm
                    gs = [p.grad for p in m.parameters() if p.grad is not None]      # mode is the optimizer's grad_scale)
                    flat = torch.cat([g.reshape(-1) for g in gs])
                    red(flat)
                    o = 0
                    for g in gs:
                        g.copy_(flat[o:o + g.numel()].view_as(g))
                        o += g.numel()

    def _own_gather(self, *optimizers):
        """Under data parallel the gradient exchange (_reduce_grads) packs a flat-buffer optimizer's per-parameter gradients and reduces them
        in place; the optimizer's step() then never gathers again -- decided here, once, so that a captured update replays what the eager
        one does (FusedAdamW.external_gather)."""
        active = self.dp is not None and (self.dp.world_size > 1 or self.dp.force_sync)
        for o in optimizers:
            if hasattr(o, "external_gather"):
                o.external_gather = bool(active)

    def _set_grad_scale(self, *optimizers):
        """sync_bn=False: the per-rank losses are means over the local batch, so the summed gradient is averaged over ranks
        (folded into the SGD kernel's gradient scale: no extra pass over the gradients)."""
        if self.dp is not None and not getattr(self.dp, "sync_bn", True):
            for o in optimizers:
                o.grad_scale = 1.0 / self.dp.world_size

    @staticmethod
    def _lr(opt):
        lr = opt.param_groups[0]["lr"]
        return 0.0 if torch.is_tensor(lr) else float(lr)      # a tensor lr (capturable AdamW) stays on the device; only FusedSGD reads this


class SupervisedStep(_StepBase):
    def __init__(self, model, args, dp=None):
        super().__init__(next(model.parameters()).device, dp)
        self.model, self.args = model, args
        self._attach(model, alone=True)
        self.optimizer = build_optimizer(args=args, model=model)
        self.lr_scheduler = build_lr_scheduler(args=args, optimizer=self.optimizer)
        self.optimizer._lr_dev = self.sc.view(S_LR1)
        self._set_grad_scale(self.optimizer)
        self._own_gather(self.optimizer)
        self.sc.host[S_COEF_A:S_COEF_A + 2] = torch.tensor([0.5, 0.5])

    def host_scalars(self, cur_itrs):
        self.sc.host[S_LR1] = self._lr(self.optimizer)

    def device_fwd_bwd(self, img, label):
        out = self.model(img)
        res = seg_loss(out, label, coef=self.sc.view(S_COEF_A, 8), dp=self.dp)
        self.optimizer.zero_grad()
        self._loss_backward(res)
        return {"loss": res[0].detach(), "logits": out.detach(), "parts": res.detach()}

    def exchange(self):
        self._reduce_grads(self.model)

    def device_update(self):
        self.optimizer.step(push_lr=False)

    def device_step(self, img, label):
        r = self.device_fwd_bwd(img, label)
        self.exchange()
        self.device_update()
        return r

    def after(self):
        self.lr_scheduler.step()

    def step(self, img, label, cur_itrs):
        self.host_scalars(cur_itrs)
        self.sc.push()
        r = self.device_step(img, label)
        self.after()
        return r


class MeanTeacherStep(_StepBase):
    def __init__(self, model, ema_model, args, dp=None):
        super().__init__(next(model.parameters()).device, dp)
        self.model, self.ema_model, self.args = model, ema_model, args
        self._attach(model, alone=True)
        self._attach(ema_model)
        self.optimizer = build_optimizer(args=args, model=model)
        self.lr_scheduler = build_lr_scheduler(args=args, optimizer=self.optimizer)
        self.optimizer._lr_dev = self.sc.view(S_LR1)
        self._set_grad_scale(self.optimizer)
        self._own_gather(self.optimizer)
        self.teacher_after = 7      # the teacher's launches enter the step behind this layer of the student's forward (see device_fwd_bwd)

    def host_scalars(self, cur_itrs, cons_w=None):
        a = self.args
        w = a.consistency * sigmoid_rampup(cur_itrs // 150, a.consistency_rampup) if cons_w is None else cons_w
        h = self.sc.host
        h[S_LR1] = self._lr(self.optimizer)
        h[S_ALPHA] = ema_alpha(cur_itrs, a.ema_decay)
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.0, 0.0, w])
        return w

    def device_fwd_bwd(self, label_img, target_label, unlabel_img):
        """Everything up to (not including) the gradient exchange."""
        nl = label_img.shape[0]
        x = cat_batch(label_img, unlabel_img)
        self._mark(0)
        if self.overlap:
            # A captured graph submits its nodes in creation order and the earlier-submitted network wins the CUs layer by layer: with the
            # teacher captured first the student -- whose forward also stores the side tensors and is what the loss waits for -- started 120 us
            # late and finished 130 - 170 us behind the teacher (in-graph stamps, tools/stream_timeline.py).  So the student's nodes come
            # first and the teacher's enter behind its 8th layer, forked from an event recorded at the step's start (no data dependency on
            # the student): both now end within ~30 us of each other, the loss starts ~60 us earlier (profiles/r04_schedule_experiments.txt).
            cur = torch.cuda.current_stream(self.dev)
            ev = torch.cuda.Event()
            ev.record(cur)
            self._mark(1)
            box = []

            def teacher():
                with torch.cuda.stream(self.side), torch.no_grad():
                    self.side.wait_event(ev)
                    self._mark(2)
                    box.append(self.ema_model(x))
                    self._mark(3)

            self.model._after_layer = (self.teacher_after, teacher)
            try:
                out = self.model(x)
            finally:          # (a forward that raises must not leave the teacher closure behind for the model's next forward, e.g. an evaluation)
                self.model._after_layer = None
            if not box:
                teacher()
            t_out = box[0]
            self._pending_join = True
        else:
            t_out = self._teacher_forward(self.ema_model, x)
            self._mark(1)
            out = self.model(x)
        self._mark(4)
        self._join_teacher(t_out)
        return self._loss_bwd(out, t_out, target_label, nl)

    def _loss_bwd(self, out, t_out, target_label, nl):
        self._mark(5)
        res = seg_loss(out, target_label, nl, coef=self.sc.view(S_COEF_A, 8), teacher_logits=t_out, dp=self.dp)
        self.optimizer.zero_grad()
        self._loss_backward(res)
        self._mark(6)
        return {"loss": res[0].detach(), "parts": res.detach(), "logits": out.detach(), "t_logits": t_out}

    def exchange(self):
        self._reduce_grads(self.model)

    def device_update(self):
        self._sgd_ema_update()
        self._mark(7)

    def device_step(self, label_img, target_label, unlabel_img):
        r = self.device_fwd_bwd(label_img, target_label, unlabel_img)
        self.exchange()
        self.device_update()
        return r

    def after(self):
        self.lr_scheduler.step()

    def step(self, label_img, target_label, unlabel_img, cur_itrs, cons_w=None):
        self.host_scalars(cur_itrs, cons_w)
        self.sc.push()
        r = self.device_step(label_img, target_label, unlabel_img)
        self.after()
        return r


def mix_samples(a: torch.Tensor, b: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    """a*(1-f[s]) + b*f[s] per sample s (ICT input mix, 2022_02_ISBI_ICT-MedSeg_ACDC.py:116-117)."""
    a, b, f = a.contiguous(), b.contiguous(), f.reshape(-1).float().contiguous()
    out = torch.empty_like(a)
    L.check(L.load().hpfg_mix_samples(L.ptr(a), L.ptr(b), L.ptr(f), L.ptr(out), a.shape[0], a[0].numel(),
                                      torch.cuda.current_stream(a.device).cuda_stream), "mix_samples")
    return out


def softmax_mix(t0: torch.Tensor, t1: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    """softmax(t0)*(1-f[s]) + softmax(t1)*f[s] of two logit tensors [n,C,H,W] -> probabilities [n,C,H,W] (ICT :126-129)."""
    x0, x1 = _nhwc(t0.detach()), _nhwc(t1.detach())
    n, H, W, Cc = x0.shape
    out = torch.empty(n, H, W, Cc, dtype=torch.float32, device=x0.device)
    L.check(L.load().hpfg_softmax_mix(L.ptr(x0), L.ptr(x1), L.ptr(f.reshape(-1).float().contiguous()), L.ptr(out), n, H, W, Cc,
                                      torch.cuda.current_stream(x0.device).cuda_stream), "softmax_mix")
    return out.permute(0, 3, 1, 2)


class ICTStep(_StepBase):
    """Interpolation consistency training (2022_02_ISBI_ICT-MedSeg_ACDC.py:110-143): the student sees
    [labelled ; mix(u0, u1)], the train-mode teacher sees u0 and u1 separately, and the consistency target is the same mix of the
    two teacher softmaxes.  Recomposition of the hot-path kernels plus two per-sample mix kernels."""

    def __init__(self, model, ema_model, args, dp=None):
        super().__init__(next(model.parameters()).device, dp)
        self.model, self.ema_model, self.args = model, ema_model, args
        self._attach(model, alone=True)
        self._attach(ema_model)
        self.optimizer = build_optimizer(args=args, model=model)
        self.lr_scheduler = build_lr_scheduler(args=args, optimizer=self.optimizer)
        self.optimizer._lr_dev = self.sc.view(S_LR1)
        self._set_grad_scale(self.optimizer)
        self._own_gather(self.optimizer)

    def draw_mix_factors(self, unlabel_bs: int, rng=None) -> torch.Tensor:
        import numpy as np
        rng = np.random if rng is None else rng
        a = float(getattr(self.args, "ict_alpha", 0.2))
        return torch.tensor(rng.beta(a, a, size=(unlabel_bs // 2, 1, 1, 1)), dtype=torch.float)      # host draw, as the reference (:111)

    def host_scalars(self, cur_itrs, cons_w=None):
        a = self.args
        w = a.consistency * sigmoid_rampup(cur_itrs // 150, a.consistency_rampup) if cons_w is None else cons_w
        h = self.sc.host
        h[S_LR1] = self._lr(self.optimizer)
        h[S_ALPHA] = ema_alpha(cur_itrs, a.ema_decay)
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.0, 0.0, w])
        return w

    def device_step(self, label_img, target_label, unlabel_img, mix_factors):
        nl, nu = label_img.shape[0], unlabel_img.shape[0]
        u0, u1 = unlabel_img[:nu // 2], unlabel_img[nu // 2:]
        f = mix_factors.to(label_img.device)
        x = torch.cat([label_img, mix_samples(u0, u1, f)], 0)
        with torch.no_grad():
            t0 = self.ema_model(u0.contiguous())
            t1 = self.ema_model(u1.contiguous())
        out = self.model(x)
        tp = softmax_mix(t0, t1, f)
        res = seg_loss(out, target_label, nl, coef=self.sc.view(S_COEF_A, 8), teacher_prob=tp, dp=self.dp)
        self.optimizer.zero_grad()
        self._loss_backward(res)
        self._reduce_grads(self.model)
        self._sgd_ema_update()
        return {"loss": res[0].detach(), "parts": res.detach(), "logits": out.detach(), "t_prob": tp}

    def after(self):
        self.lr_scheduler.step()

    def step(self, label_img, target_label, unlabel_img, cur_itrs, mix_factors=None, cons_w=None):
        if mix_factors is None:
            mix_factors = self.draw_mix_factors(unlabel_img.shape[0])
        self.host_scalars(cur_itrs, cons_w)
        self.sc.push()
        r = self.device_step(label_img, target_label, unlabel_img, mix_factors)
        self.after()
        return r


def noise_add(x: torch.Tensor, noise: torch.Tensor, scale: float = 0.1, lo: float = -0.2, hi: float = 0.2) -> torch.Tensor:
    """x.repeat(k,1,1,1) + clamp(noise*scale, lo, hi) with k = noise.shape[0] // x.shape[0] (UAMT teacher inputs,
    2019_07_MICCAI_Uncertainty_Aware_ACDC.py:130-131,137-142)."""
    x, noise = x.contiguous(), noise.float().contiguous()
    if noise.numel() % x.numel():
        raise ValueError("noise must hold a whole number of copies of x")
    out = torch.empty_like(noise)
    L.check(L.load().hpfg_noise_add(L.ptr(x), L.ptr(noise), L.ptr(out), x.numel(), noise.numel(), scale, lo, hi,
                                    torch.cuda.current_stream(x.device).cuda_stream), "noise_add")
    return out


def uncertainty_mask(pred_blocks, n_images: int, threshold_dev: torch.Tensor, want_uncertainty: bool = False):
    """Entropy mask of UAMT (:144-151,162-163).  pred_blocks: teacher logits [k*n_images,C,H,W] each, stacked in the order the
    reference fills `preds`; prediction t of image s is row t*n_images + s.  Returns mask [n_images,1,H,W] (and the entropy)."""
    xs = [_nhwc(b.detach().float()) for b in pred_blocks]
    per, H, W, Cc = xs[0].shape
    if any(tuple(x.shape) != (per, H, W, Cc) for x in xs) or (len(xs) * per) % n_images:
        raise ValueError("prediction blocks must have one shape and hold T*n_images predictions")
    T = len(xs) * per // n_images
    pb = L.PredBlocks()
    for i, x in enumerate(xs):
        pb.p[i] = x.data_ptr()
    pb.n_blocks, pb.per_block = len(xs), per
    dev = xs[0].device
    mask = torch.empty(n_images, 1, H, W, dtype=torch.float32, device=dev)
    unc = torch.empty(n_images, 1, H, W, dtype=torch.float32, device=dev) if want_uncertainty else None
    L.check(L.load().hpfg_uncertainty_mask(C.byref(pb), T, n_images, H, W, Cc, L.ptr(threshold_dev), L.ptr(mask), L.ptr(unc),
                                           torch.cuda.current_stream(dev).cuda_stream), "uncertainty_mask")
    return (mask, unc) if want_uncertainty else mask


class UAMTStep(_StepBase):
    """Uncertainty-aware Mean Teacher (2019_07_MICCAI_Uncertainty_Aware_ACDC.py:110-170): student forward on
    [labelled ; unlabelled], one noisy teacher forward for the consistency target, T=8 further noisy teacher predictions
    (T/2 forwards of the doubled unlabelled batch, all in train mode like the reference), entropy of their mean softmax -> mask,
    masked softmax-MSE + 0.5*(CE + Dice), SGD, EMA.  Recomposition of the hot-path kernels plus noise / entropy-mask kernels."""

    T = 8

    def __init__(self, model, ema_model, args, dp=None):
        super().__init__(next(model.parameters()).device, dp)
        self.model, self.ema_model, self.args = model, ema_model, args
        self._attach(model, alone=True)
        self._attach(ema_model)
        self.optimizer = build_optimizer(args=args, model=model)
        self.lr_scheduler = build_lr_scheduler(args=args, optimizer=self.optimizer)
        self.optimizer._lr_dev = self.sc.view(S_LR1)
        self._set_grad_scale(self.optimizer)
        self._own_gather(self.optimizer)

    def host_scalars(self, cur_itrs, cons_w=None):
        import math
        a = self.args
        w = a.consistency * sigmoid_rampup(cur_itrs // 150, a.consistency_rampup) if cons_w is None else cons_w
        h = self.sc.host
        h[S_LR1] = self._lr(self.optimizer)
        h[S_ALPHA] = ema_alpha(cur_itrs, a.ema_decay)
        h[S_THRESH] = (0.75 + 0.25 * sigmoid_rampup(cur_itrs, a.total_itrs)) * math.log(2)          # :162
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.0, 0.0, w])
        return w

    def draw_noise(self, unlabel_img):
        """The reference's draws, in its order (:130, :142): one [Nu] field, then T/2 fields of [2*Nu] (torch device generator)."""
        n0 = torch.randn_like(unlabel_img)
        rest = [torch.randn(2 * unlabel_img.shape[0], *unlabel_img.shape[1:], device=unlabel_img.device) for _ in range(self.T // 2)]
        return n0, rest

    def device_step(self, label_img, target_label, unlabel_img, noise0, noises):
        nl, nu = label_img.shape[0], unlabel_img.shape[0]
        out = self.model(cat_batch(label_img, unlabel_img))
        with torch.no_grad():
            ema_out = self.ema_model(noise_add(unlabel_img, noise0))
            preds = [self.ema_model(noise_add(unlabel_img, nz)) for nz in noises]
        mask = uncertainty_mask(preds, nu, self.sc.view(S_THRESH))
        res = seg_loss(out, target_label, nl, coef=self.sc.view(S_COEF_A, 8), teacher_logits=ema_out, cons_mask=mask, dp=self.dp)
        self.optimizer.zero_grad()
        self._loss_backward(res)
        self._reduce_grads(self.model)
        self._sgd_ema_update()
        return {"loss": res[0].detach(), "parts": res.detach(), "logits": out.detach(), "mask": mask, "t_logits": ema_out.detach()}

    def after(self):
        self.lr_scheduler.step()

    def step(self, label_img, target_label, unlabel_img, cur_itrs, noise=None, cons_w=None):
        n0, rest = self.draw_noise(unlabel_img) if noise is None else noise
        self.host_scalars(cur_itrs, cons_w)
        self.sc.push()
        r = self.device_step(label_img, target_label, unlabel_img, n0, rest)
        self.after()
        return r


class CPSStep(_StepBase):
    def __init__(self, model1, model2, args, dp=None):
        super().__init__(next(model1.parameters()).device, dp)
        self.model1, self.model2, self.args = model1, model2, args
        self._attach(model1)
        self._attach(model2)
        self.optimizer1 = build_optimizer(args=args.model1, model=model1)
        self.optimizer2 = build_optimizer(args=args.model2, model=model2)
        self.lr_scheduler1 = build_lr_scheduler(args=args.model1, optimizer=self.optimizer1)
        self.lr_scheduler2 = build_lr_scheduler(args=args.model2, optimizer=self.optimizer2)
        self.optimizer1._lr_dev = self.sc.view(S_LR1)
        self.optimizer2._lr_dev = self.sc.view(S_LR2)
        self._set_grad_scale(self.optimizer1, self.optimizer2)
        self._own_gather(self.optimizer1, self.optimizer2)

    def host_scalars(self, cur_itrs, cons_w=None):
        a = self.args
        w = a.consistency * sigmoid_rampup(cur_itrs // 150, a.consistency_rampup) if cons_w is None else cons_w
        h = self.sc.host
        h[S_LR1], h[S_LR2] = self._lr(self.optimizer1), self._lr(self.optimizer2)
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.5 * w, 0.5 * w, 0.0])
        return w

    def device_fwd_bwd(self, label_img, target_label, unlabel_img):
        """Everything up to (not including) the gradient exchange."""
        nl = label_img.shape[0]
        x = cat_batch(label_img, unlabel_img)
        if self.overlap:
            # the two students are independent until the losses: the second one's forward runs on the side stream, and autograd runs its
            # backward there too (a backward node executes on the stream of its forward), so both chains of small kernels overlap
            cur = torch.cuda.current_stream(self.dev)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                o2 = self.model2(x)
            o1 = self.model1(x)
            cur.wait_stream(self.side)
            o2.record_stream(cur)
        else:
            o1 = self.model1(x)
            o2 = self.model2(x)
        p1 = argmax_labels(o1[nl:])
        p2 = argmax_labels(o2[nl:])
        coef = self.sc.view(S_COEF_A, 8)
        r1 = seg_loss(o1, target_label, nl, coef=coef, pseudo=p2, dp=self.dp)
        r2 = seg_loss(o2, target_label, nl, coef=coef, pseudo=p1, dp=self.dp)
        loss = r1[0] + r2[0]
        self.optimizer1.zero_grad()
        self.optimizer2.zero_grad()
        loss.backward()
        self._join_backward(self.side)
        return {"loss": loss.detach(), "parts1": r1.detach(), "parts2": r2.detach(), "logits1": o1.detach(), "logits2": o2.detach()}

    def exchange(self):
        self._reduce_grads(self.model1, self.model2)

    def device_update(self):
        for o in (self.optimizer1, self.optimizer2):
            o.step(push_lr=False) if hasattr(o, "push_lr") else o.step()         # FusedSGD reads its lr from the device scalars

    def device_step(self, label_img, target_label, unlabel_img):
        r = self.device_fwd_bwd(label_img, target_label, unlabel_img)
        self.exchange()
        self.device_update()
        return r

    def after(self):
        self.lr_scheduler1.step()
        self.lr_scheduler2.step()

    def step(self, label_img, target_label, unlabel_img, cur_itrs, cons_w=None):
        self.host_scalars(cur_itrs, cons_w)
        self.sc.push()
        r = self.device_step(label_img, target_label, unlabel_img)
        self.after()
        return r


class HPFGStep(_StepBase):
    def __init__(self, model1, model2, ema_model, args, dp=None):
        from .utils import Dense_Loss
        super().__init__(next(model1.parameters()).device, dp)
        self.model1, self.model2, self.ema_model, self.args = model1, model2, ema_model, args
        for m in (model1, model2, ema_model):
            self._attach(m)
        self.optimizer1 = build_optimizer(args=args.model1, model=model1)
        self.optimizer2 = build_optimizer(args=args.model2, model=model2)
        self.lr_scheduler1 = build_lr_scheduler(args=args.model1, optimizer=self.optimizer1)
        self.lr_scheduler2 = build_lr_scheduler(args=args.model2, optimizer=self.optimizer2)
        self.optimizer1._lr_dev = self.sc.view(S_LR1)
        self.optimizer2._lr_dev = self.sc.view(S_LR2)
        self._set_grad_scale(self.optimizer1, self.optimizer2)
        self._own_gather(self.optimizer1, self.optimizer2)
        if hasattr(model1, "dense_projection_high"):
            # main.py:152 discards the first student's neck outputs: their parameters never get a gradient (torch's SGD then skips them: no
            # weight decay, no momentum), so the necks are not computed at all and the optimizer stops at the backbone
            model1.skip_necks = True
            if isinstance(self.optimizer1, FusedSGD):
                self.optimizer1.active_numel = model1._backbone_numel
        self.dense_loss = Dense_Loss(args.batch_size + args.unlabel_batch_size, self.dev)
        self.dense_loss.dp = dp      # global-batch mode: NT-Xent over the gathered features of all ranks (main.py:172 contrasts the whole batch)
        self.mask_generator = BoxMaskGenerator(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True,
                                               within_bounds=True, invert=True)          # main.py:94-115
        self._w = 0.0

    def host_scalars(self, cur_itrs):
        a = self.args
        w = a.consistency * linear_rampup(cur_itrs // 150, a.consistency_rampup)
        h = self.sc.host
        h[S_LR1], h[S_LR2] = self._lr(self.optimizer1), self._lr(self.optimizer2)
        h[S_ALPHA] = ema_alpha(cur_itrs, a.ema_decay)
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.0, 7.0 * w, 0.0])                 # model1: sup + 7w * pseudo Dice
        h[S_COEF_B:S_COEF_B + 5] = torch.tensor([0.5, 0.5, 0.0, 0.0, 0.0 if cur_itrs < 1000 else w])   # model2: sup + w * MSE
        h[3] = w
        self._w = w
        return w

    def make_cutmix_mask(self, n, shape, rng=None, device=None):
        """Host numpy masks (reference behaviour) or, with ``device``, the same masks rasterised on the GPU."""
        if device is not None:
            return self.mask_generator.generate_params_device(n_masks=n, mask_shape=shape, device=device, rng=rng)
        m = self.mask_generator.generate_params(n_masks=n, mask_shape=shape, rng=rng)
        return torch.tensor(m, dtype=torch.float)

    def device_fwd_bwd(self, label_img, target_label, label_img1, target_label1, img_unlabel, cutmix_mask):
        """label_img1/target_label1 are already repeated to the unlabelled batch size (main.py:142-143)."""
        nl = label_img.shape[0]
        mix_un = cutmix_blend(label_img1, img_unlabel, cutmix_mask)
        batch_mix = torch.cat([label_img, mix_un], 0)
        split = self.overlap
        if split:      # student 1 (fed the CutMix batch) is independent of student 2 and the teacher until the losses: a stream of its own,
            # forward and -- through autograd, which runs a backward node on the stream of its forward -- backward (see CPSStep)
            cur = torch.cuda.current_stream(self.dev)
            if getattr(self, "side2", None) is None:
                self.side2 = torch.cuda.Stream(device=self.dev)
            self.side2.wait_stream(cur)
            with torch.cuda.stream(self.side2):
                o1, _, _ = self.model1(batch_mix)
        else:
            o1, _, _ = self.model1(batch_mix)
        volume = cat_batch(label_img, img_unlabel)
        volume_t = volume
        ot, th1, th2 = self._teacher_forward(self.ema_model, volume_t)
        o2, h1, h2 = self.model2(volume)
        self._join_teacher(ot, th1, th2)
        if split:
            cur.wait_stream(self.side2)
            o1.record_stream(cur)
            batch_mix.record_stream(self.side2)
        pseudo = argmax_labels(ot[nl:], target_label1, cutmix_mask[:, 0])
        r1 = seg_loss(o1, target_label, nl, coef=self.sc.view(S_COEF_A, 8), pseudo=pseudo, dp=self.dp)
        r2 = seg_loss(o2, target_label, nl, coef=self.sc.view(S_COEF_B, 8), teacher_logits=ot, dp=self.dp)
        contrast = self.dense_loss(h1, th1) + self.dense_loss(h2, th2)
        loss = r1[0] + r2[0] + self.sc.view(3)[0] * contrast
        self.optimizer1.zero_grad()
        self.optimizer2.zero_grad()
        loss.backward()
        self._join_backward(getattr(self, "side2", None))
        return {"loss": loss.detach(), "parts1": r1.detach(), "parts2": r2.detach(), "contrast": contrast.detach(),
                "logits1": o1.detach(), "logits2": o2.detach(), "t_logits": ot}

    def exchange(self):
        self._reduce_grads(self.model1, self.model2)

    def device_update(self):
        self.optimizer1.step(push_lr=False)
        self.optimizer2.step(push_lr=False)
        a = self.sc.view(S_ALPHA)
        update_ema_variables_backbone(self.model1, self.model2, self.args.ema_decay, 0, alpha_dev=a)
        update_ema_variables(self.model2, self.ema_model, self.args.ema_decay, 0, alpha_dev=a)

    def device_step(self, label_img, target_label, label_img1, target_label1, img_unlabel, cutmix_mask):
        r = self.device_fwd_bwd(label_img, target_label, label_img1, target_label1, img_unlabel, cutmix_mask)
        self.exchange()
        self.device_update()
        return r

    def after(self):
        self.lr_scheduler1.step()
        self.lr_scheduler2.step()

    def step(self, label_img, target_label, label_img1, target_label1, img_unlabel, cutmix_mask, cur_itrs):
        self.host_scalars(cur_itrs)
        self.sc.push()
        r = self.device_step(label_img, target_label, label_img1, target_label1, img_unlabel, cutmix_mask)
        self.after()
        return r


class S4CVNetStep(_StepBase):
    """S4CVnet (2022_08_CVPR_S4CVNet_ACDC.py:107-167): two students on [labelled ; unlabelled], the EMA teacher
    of model2 on the unlabelled images + clamp(N(0,1) * 0.1, +-0.2) (on its own stream), cross Dice pseudo supervision weighted 7w, and from
    iteration 1000 on the softmax MSE of each student against the teacher weighted w (linear ramp-up), SGD on both, EMA(model2 -> teacher).
    Pure recomposition of the hot-path kernels (noise_add, argmax_labels, the fused loss with an unlabelled-only consistency target)."""

    def __init__(self, model1, model2, ema_model, args, dp=None):
        super().__init__(next(model1.parameters()).device, dp)
        self.model1, self.model2, self.ema_model, self.args = model1, model2, ema_model, args
        for m in (model1, model2, ema_model):
            self._attach(m)
        self.optimizer1 = build_optimizer(args=args.model1, model=model1)
        self.optimizer2 = build_optimizer(args=args.model2, model=model2)
        self.lr_scheduler1 = build_lr_scheduler(args=args.model1, optimizer=self.optimizer1)
        self.lr_scheduler2 = build_lr_scheduler(args=args.model2, optimizer=self.optimizer2)
        self.optimizer1._lr_dev = self.sc.view(S_LR1)
        self.optimizer2._lr_dev = self.sc.view(S_LR2)
        self._set_grad_scale(self.optimizer1, self.optimizer2)
        self._own_gather(self.optimizer1, self.optimizer2)

    def host_scalars(self, cur_itrs):
        a = self.args
        w = a.consistency * linear_rampup(cur_itrs // 150, a.consistency_rampup)
        h = self.sc.host
        h[S_LR1], h[S_LR2] = self._lr(self.optimizer1), self._lr(self.optimizer2)
        h[S_ALPHA] = ema_alpha(cur_itrs, a.ema_decay)
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.0, 7.0 * w, 0.0 if cur_itrs < 1000 else w])      # both students (:141-150)
        return w

    def draw_noise(self, unlabel_img):
        return torch.randn_like(unlabel_img)          # :109, clamped inside noise_add

    def device_fwd_bwd(self, label_img, target_label, unlabel_img, noise):
        nl = label_img.shape[0]
        x = cat_batch(label_img, unlabel_img)
        ot = self._teacher_forward(self.ema_model, noise_add(unlabel_img, noise))
        if self.overlap:      # the U-Net on a stream of its own, forward and backward (see CPSStep)
            cur = torch.cuda.current_stream(self.dev)
            if getattr(self, "side2", None) is None:
                self.side2 = torch.cuda.Stream(device=self.dev)
            self.side2.wait_stream(cur)
            with torch.cuda.stream(self.side2):
                o1 = self.model1(x)
            o2 = self.model2(x)
            cur.wait_stream(self.side2)
            o1.record_stream(cur)
        else:
            o1 = self.model1(x)
            o2 = self.model2(x)
        self._join_teacher(ot)
        p1 = argmax_labels(o1[nl:])
        p2 = argmax_labels(o2[nl:])
        coef = self.sc.view(S_COEF_A, 8)
        r1 = seg_loss(o1, target_label, nl, coef=coef, pseudo=p2, teacher_logits=ot, dp=self.dp)
        r2 = seg_loss(o2, target_label, nl, coef=coef, pseudo=p1, teacher_logits=ot, dp=self.dp)
        loss = r1[0] + r2[0]
        self.optimizer1.zero_grad()
        self.optimizer2.zero_grad()
        loss.backward()
        self._join_backward(getattr(self, "side2", None))
        return {"loss": loss.detach(), "parts1": r1.detach(), "parts2": r2.detach(), "logits1": o1.detach(), "logits2": o2.detach(), "t_logits": ot}

    def exchange(self):
        self._reduce_grads(self.model1, self.model2)

    def device_update(self):
        self.optimizer1.step(push_lr=False)
        self.optimizer2.step(push_lr=False)
        update_ema_variables(self.model2, self.ema_model, self.args.ema_decay, 0, alpha_dev=self.sc.view(S_ALPHA))

    def device_step(self, label_img, target_label, unlabel_img, noise):
        r = self.device_fwd_bwd(label_img, target_label, unlabel_img, noise)
        self.exchange()
        self.device_update()
        return r

    def after(self):
        self.lr_scheduler1.step()
        self.lr_scheduler2.step()

    def step(self, label_img, target_label, unlabel_img, cur_itrs, noise=None):
        if noise is None:
            noise = self.draw_noise(unlabel_img)
        self.host_scalars(cur_itrs)
        self.sc.push()
        r = self.device_step(label_img, target_label, unlabel_img, noise)
        self.after()
        return r


class GraphNotCapturable(RuntimeError):
    """GraphedStep's refusal BEFORE anything is captured (a step with host-launched collectives between its kernels): the one condition a
    driver loop answers by staying eager.  Anything raised DURING a capture (an argument check, a workspace that is too small) is a real
    failure and propagates."""


class GraphedStep:
    """Captures ``step_obj.device_step`` on static input buffers into one hipGraph (torch.cuda.CUDAGraph) and replays it.
    Host scalars are copied to the device eagerly in front of every replay (StepScalars ring); dropout masks change per replay
    through the engines' device seed word."""

    def __init__(self, step_obj, example_inputs, warmup: int = 3, alias_inputs: bool = False, before_capture=None):
        """alias_inputs: the graph reads ``example_inputs`` themselves (the caller refills those tensors in place, or they never
        change) instead of private copies that every step() would have to refresh with one copy kernel per input.
        before_capture: optional callable run between the eager warm-up steps and the capture (bench.py resets its time-stamp logs there)."""
        self.s = step_obj
        dp0 = getattr(step_obj, "dp", None)
        # (HPFG's Dense_Loss gathers the neck features of all ranks: through the peer windows when they are enabled -- kernels, capturable --
        # otherwise with a host-launched collective)
        host_gather = isinstance(step_obj, HPFGStep) and not getattr(dp0, "p2p_grads", False)
        if dp0 is not None and getattr(dp0, "sync_bn", True) and (dp0.world_size > 1 or dp0.force_sync) and (not getattr(dp0, "p2p", False) or host_gather):
            # all-reduced BatchNorm statistics = a collective between the kernels of every layer: those never sit inside a captured region
            # (RCCL's watchdog thread polls its events while a capture is open; losing that race aborted the process)
            raise GraphNotCapturable("GraphedStep: a step with collectives between its kernels (data parallel with sync_bn=True) is not captured into a "
                               "hipGraph; run it eager, enable the peer mailbox exchange (DataParallelContext.enable_peer_exchange: the sums then cross "
                               "the ranks inside the kernels), or use sync_bn=False (per-rank BatchNorm: collectives only between graphs)")
        self.alias = bool(alias_inputs)
        self.static = list(example_inputs) if self.alias else [t.clone() for t in example_inputs]
        self.graph = torch.cuda.CUDAGraph()
        self._seed_words = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):
                self.s.host_scalars(i + 1)
                self.s.sc.push()
                self.s.device_step(*self.static)
                self.s.after()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if before_capture is not None:
            before_capture()
        if dp0 is not None:
            dp0._loss_call = 0          # the capture traces one step from its beginning
        # thread_local: another thread's HIP calls (the RCCL watchdog polling its events) must not invalidate this capture
        dp = getattr(step_obj, "dp", None)
        # data parallel: the gradient exchange is never captured -- [forward + loss + backward] | eager all-reduce(s) | [update] as a chain of
        # graphs, in the per-rank BatchNorm mode and (BatchNorm / loss sums exchanged by the kernels through peer mailboxes) the global-batch mode
        # (with the peer-window gradient exchange there is no host-launched collective left: one graph, as on one GPU)
        self.split = bool(dp is not None and (dp.world_size > 1 or dp.force_sync) and not getattr(dp, "p2p_grads", False))
        if self.split and not hasattr(step_obj, "device_fwd_bwd"):
            raise GraphNotCapturable(f"GraphedStep: {type(step_obj).__name__} has no device_fwd_bwd / exchange / device_update split, so its gradient "
                               "exchange would be captured into the graph; run it eager under data parallel")
        self._freeze_seed_updates(True)
        try:
            self._capture(dp, mode_split=self.split)
        finally:          # whatever a capture raises, the networks leave graph-seed mode and no bucket hook stays installed
            self._freeze_seed_updates(False)
            if dp is not None:
                dp.bucket_hook = None

    def _capture(self, dp, mode_split: bool):
        if self.split:
            # per-rank BatchNorm (DDP semantics): the only collectives of the step are the gradient all-reduces.  The work between
            # them is captured as a CHAIN of hipGraphs and the RCCL calls are issued eagerly in between, on a side stream -- no
            # collective node inside a hipGraph: [forward + loss + decoder half of backward] -> bucket 0 || [encoder half] -> bucket 1
            # -> [SGD + EMA].  A bucket boundary inside backward (DataParallelContext.launch_bucket -> bucket_hook) ends the graph
            # being captured and begins the next one in the same memory pool.
            self.graph_b = torch.cuda.CUDAGraph()
            self.graphs, self.bucket_after = [self.graph], []
            # the bucket chain ends one graph and begins the next in the middle of backward: only legal while no second stream is forked
            # there, i.e. for steps with one trainable network (two students back-propagate on two streams: one exchange after both)
            overlap = (bool(getattr(dp, "overlap", False))
                       and sum(1 for m in self._models() if any(p.requires_grad for p in m.parameters())) == 1)
            if overlap:
                # the boundary runs on autograd's device thread: ending a capture from another thread than the one that began it
                # needs the relaxed mode (which also tolerates the RCCL watchdog's event queries)
                mode = "relaxed"
                pad = torch.zeros(1, device=self.static[0].device)
                cap = torch.cuda.Stream()
                torch.cuda.synchronize()
                cap.wait_stream(torch.cuda.current_stream())

                def hook(t):
                    self.graphs[-1].capture_end()
                    self.bucket_after.append(t)
                    g2 = torch.cuda.CUDAGraph()
                    g2.capture_begin(pool=self.graph.pool(), capture_error_mode=mode)
                    self.graphs.append(g2)
                    pad.add_(1.0)          # no graph of the chain is empty (the last boundary is the end of backward)

                with torch.cuda.stream(cap):
                    self.graph.capture_begin(capture_error_mode=mode)
                    dp.bucket_hook = hook
                    try:
                        self.out = self.s.device_fwd_bwd(*self.static)
                    finally:
                        dp.bucket_hook = None
                        self.graphs[-1].capture_end()
                torch.cuda.current_stream().wait_stream(cap)
                for m in self._models():
                    m._buckets_launched = False
            else:
                prev, dp.overlap = getattr(dp, "overlap", False), False      # one blocking exchange between the two graphs
                try:
                    with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                        self.out = self.s.device_fwd_bwd(*self.static)
                finally:
                    dp.overlap = prev
            with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), capture_error_mode="thread_local"):
                self.s.device_update()
        else:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = self.s.device_step(*self.static)

    def _models(self):
        return [getattr(self.s, n) for n in ("model", "model1", "model2", "ema_model") if hasattr(self.s, n)]

    def _freeze_seed_updates(self, capturing: bool):
        for m in self._models():
            m._graph_seed_mode = capturing
            if capturing:
                m._graph_fwds = 0
            else:
                m._fwds_per_replay = getattr(m, "_graph_fwds", 0)      # (UNet.bump_graph_seed keeps the host's seed counter in step with the replays)

    def step(self, inputs, cur_itrs, **kw):
        for dst, src in zip(self.static, inputs):
            if dst is not src:
                dst.copy_(src, non_blocking=_async_ok(src))
        self.s.host_scalars(cur_itrs, **kw)
        self.s.sc.push()          # eager H2D copy of this step's scalars from a fresh ring slot, ordered in front of the replay
        for m in self._models():
            m.bump_graph_seed()      # (the U-Nets advance their seed words inside the captured forward; the SegFormer branch does it here)
        if self.split:
            dp = self.s.dp
            for i, g in enumerate(self.graphs):
                g.replay()
                if i < len(self.bucket_after):
                    dp.launch_bucket(self.bucket_after[i])      # eager RCCL call on the side stream; the next graph runs beside it
            if self.bucket_after:
                dp.join_buckets()
            else:
                self.s.exchange()
            self.graph_b.replay()
        else:
            self.graph.replay()
        self.s.after()
        return self.out


class CTCTStep(CPSStep):
    """Cross teaching between a CNN and a transformer (2021_12_MIDL_CTCT_ACDC.py:117-134): model1 (U-Net, HIP
    engine, FusedSGD) and model2 (SegFormer, AdamW) see [labelled ; unlabelled]; each is supervised by 0.5*(CE + Dice) on the labelled
    part and by w * Dice against the OTHER network's arg-max on the unlabelled part.  Same kernels as the CPS step; only the loss
    weights differ (Dice-only pseudo-supervision, weight w instead of 0.5*w*(CE + Dice))."""

    def host_scalars(self, cur_itrs, cons_w=None):
        a = self.args
        w = a.consistency * sigmoid_rampup(cur_itrs // 150, a.consistency_rampup) if cons_w is None else cons_w
        h = self.sc.host
        h[S_LR1], h[S_LR2] = self._lr(self.optimizer1), self._lr(self.optimizer2)
        h[S_COEF_A:S_COEF_A + 5] = torch.tensor([0.5, 0.5, 0.0, w, 0.0])
        return w


# ------------------------------------------------------------------------------------------------------------------------
# Driver loops with the reference's names / signatures: same iteration law (two labelled iterators restarted on StopIteration,
# return once cur_itrs > total_itrs), evaluation every ``step_size`` iterations through hpfg_amd.val.test_acdc, and best-Dice
# checkpoints in the reference's dict format {"model", "optimizer", "lr_scheduler", "cur_itrs", "best_dice"}.
#
# Every loop runs the path bench.py times: iteration 1 is an eager step (it allocates the engines' workspaces and IS iteration 1),
# iteration 2 captures the step into ONE hipGraph on static input buffers, every later iteration is one copy per input (host -> the static
# buffer) + one replay (``_LoopRunner``; ``args.hipgraph = False`` or HPFG_LOOP_GRAPH=0 keeps eager launches).  The reference's
# per-iteration ``writer.add_scalar`` calls (main.py:216-222, 2017_03...py:111-113, sup_ACDC.py:96-97, 2021_06...py:124-128,
# 2021_12...py:158-161) are served by ``ScalarLog``: the loss vectors stay on the device in a ring and reach ``args.writer`` (if there is
# one) every ``args.log_every`` iterations with ONE device-to-host copy -- the scalar NAMES and values are the reference's, the
# ``loss.item()`` synchronisation per iteration is not.  tqdm output is not produced.
# ------------------------------------------------------------------------------------------------------------------------
def _cycle(loader):
    it = iter(loader)
    while True:
        try:
            yield next(it)
        except StopIteration:
            it = iter(loader)
            yield next(it)


class ScalarLog:
    """Per-iteration scalars without a host synchronisation per iteration.  ``add(itr, vecs, host)`` parks the device vectors ``vecs`` (the
    fused-loss outputs [total, ce0, dice0, ce1, dice1, mse, 0, 0] of each network, ...) in row ``k`` of a device ring with one small copy
    each and remembers the host-side scalars (learning rates, consistency weights) of that iteration; every ``every`` iterations -- and at
    ``flush()`` -- ONE device-to-host copy brings the rows back and ``emit(row, host) -> {name: value}`` names them for
    ``writer.add_scalar(name, value, itr)``.  ``history`` keeps (itr, {name: value}) of everything flushed; ``losses()`` = the per-iteration
    total loss (row element 0 unless ``emit`` names a "<prefix>/loss")."""

    def __init__(self, dev, width: int, emit, writer=None, every: int = 50):
        self.dev, self.width, self.emit, self.writer = dev, int(width), emit, writer
        self.every = max(1, int(every))
        self.buf = torch.zeros(self.every, self.width, dtype=torch.float32, device=dev)
        self.pending, self.history = [], []
        self.check = None          # optional callable run with every flush (the driver loops: DataParallelContext.check_peer_errors)

    def add(self, itr: int, vecs, host: Dict[str, float]):
        k, o = len(self.pending), 0
        for v in vecs:
            v = v.detach().reshape(-1)
            self.buf[k, o:o + v.numel()].copy_(v, non_blocking=True)
            o += v.numel()
        assert o <= self.width
        self.pending.append((int(itr), dict(host)))
        if len(self.pending) == self.every:
            self.flush()

    def flush(self):
        if not self.pending:
            return
        rows = self.buf[:len(self.pending)].cpu()          # the one synchronisation per `every` iterations
        if self.check is not None:          # data parallel: the peer-exchange error word, read while the device is idle anyway
            self.check()
        for (itr, host), row in zip(self.pending, rows):
            named = self.emit([float(x) for x in row], host)
            self.history.append((itr, named))
            if self.writer is not None:
                for name, val in named.items():
                    self.writer.add_scalar(name, val, itr)
        self.pending = []

    def losses(self) -> torch.Tensor:
        self.flush()
        vals = [next((v for n, v in named.items() if n.endswith("/loss")), float("nan")) for _, named in self.history]
        return torch.tensor(vals, dtype=torch.float32, device=self.dev)


def _async_ok(src: torch.Tensor) -> bool:
    """May a copy of `src` to the device be queued without waiting for it?  Device tensors and pinned host batches (what a DataLoader with
    pin_memory hands over): yes.  A pageable host tensor -- e.g. the handful of per-iteration draws a driver makes with numpy -- is a
    temporary whose memory may be reused before a queued copy has read it: copy it before returning."""
    return src.is_cuda or src.is_pinned()


class _LoopRunner:
    """One training iteration of a driver loop: eager for iteration 1, captured at iteration 2, replayed afterwards (see the section header)."""

    def __init__(self, st, args, slog: ScalarLog, vec_keys, host_fn):
        self.st, self.args, self.slog, self.vec_keys, self.host_fn = st, args, slog, vec_keys, host_fn
        self.graph_ok = bool(getattr(args, "hipgraph", True)) and os.environ.get("HPFG_LOOP_GRAPH", "1") == "1"
        self.runner, self.n = None, 0
        self.dev = torch.device(args.device)
        dp = getattr(args, "dp", None)
        if dp is not None and slog.check is None:          # a poll that expired means partial sums: noticed within `log_every` iterations
            slog.check = dp.check_peer_errors

    def _to_dev(self, t):
        t = t.to(self.dev, non_blocking=_async_ok(t))
        return t.float() if t.is_floating_point() and t.dtype != torch.float32 else t

    def __call__(self, inputs, cur_itrs: int, **kw):
        """inputs: the step's tensor arguments as the loaders hand them over (host or device); images are converted to fp32 as the
        reference's ``.to(args.device).float()`` does."""
        self.n += 1
        r = None
        if self.graph_ok and self.n >= 2:
            if self.runner is None:
                dev_in = [self._to_dev(t) for t in inputs]
                try:
                    self.runner = GraphedStep(self.st, dev_in, warmup=0)          # captures; executes nothing
                except GraphNotCapturable as e:          # collectives between the step's kernels: stay eager (any other error propagates)
                    logger = getattr(self.args, "logger", None)
                    if logger is not None:
                        logger.warning(f"hipGraph capture unavailable ({e}); running eager")
                    self.graph_ok = False
            if self.runner is not None and all(tuple(a.shape) == tuple(b.shape) for a, b in zip(inputs, self.runner.static)):
                r = self.runner.step(list(inputs), cur_itrs, **kw)          # copies (and converts) each input into its static buffer, replays
        if r is None:          # the eager form of the same iteration (what every step object's .step() does)
            self.st.host_scalars(cur_itrs, **kw)
            self.st.sc.push()
            r = self.st.device_step(*[self._to_dev(t) for t in inputs])
            self.st.after()
        self.slog.add(cur_itrs, [r[k] for k in self.vec_keys], self.host_fn())
        return r


def _writer(args):
    return getattr(args, "writer", None)


def _log_every(args):
    return int(getattr(args, "log_every", 50) or 50)


def _sup_of(row, o=0):
    """0.5 * (CE + Dice) on the labelled images (the supervised term of every driver: coefficients 0.5 / 0.5) of the loss vector at offset o."""
    return 0.5 * (row[o + 1] + row[o + 2])


class _Best:
    """Periodic evaluation + best-Dice checkpoint of one network (main.py:224-279, 2017_03...py:116-152, sup_ACDC.py:97-116)."""

    def __init__(self, args, key, path_attr=None, path_fmt=None):
        self.args, self.key, self.best = args, key, 0.0
        self.path_attr, self.path_fmt = path_attr, path_fmt

    def path(self):
        if self.path_fmt is not None and getattr(self.args, "save_path", None):
            return os.path.join(self.args.save_path, "model", self.path_fmt.format(self.best))
        return getattr(self.args, self.path_attr, None) if self.path_attr else None

    def __call__(self, model, optimizer, lr_scheduler, test_loader, cur_itrs, name="test"):
        from .val import test_acdc
        dice, hd95 = test_acdc(model=model, test_loader=test_loader, args=self.args, cur_itrs=cur_itrs, name=name)
        logger = getattr(self.args, "logger", None)
        if logger is not None:
            logger.info("{}_dice: {:.4f} {}_hd95: {:.4f}".format(self.key, dice, self.key, hd95))
        if dice > self.best:
            self.best = dice
            path = self.path()
            if path:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict(),
                            "cur_itrs": cur_itrs, "best_dice": self.best}, path)
        model.train()
        return dice


def _due(cur_itrs, args, test_loader):
    return test_loader is not None and cur_itrs % args.step_size == 0


def _check_peers(args, cur_itrs, final=False):
    """Data parallel with the peer exchanges: a kernel whose poll for a peer's value expired carried on with partial sums (it must not hang
    the GPU) and set a device word -- stop training on it at the evaluation cadence and at the end, not silently update weights from it."""
    dp = getattr(args, "dp", None)
    if dp is not None and (final or cur_itrs % args.step_size == 0):
        dp.check_peer_errors()


def Supervise(model, train_loader, test_loader, args):
    """sup_ACDC.py:59-125."""
    st = SupervisedStep(model, args, getattr(args, "dp", None))
    best = _Best(args, "model", path_fmt="model_{:.4f}.pth")
    model.train()
    slog = ScalarLog(torch.device(args.device), 8, lambda row, h: {"supervise/loss": row[0], "supervise/lr": h["lr"]}, _writer(args), _log_every(args))
    run = _LoopRunner(st, args, slog, ["parts"], lambda: {"lr": st._lr(st.optimizer)})
    cur_itrs = 0
    max_epoch = args.total_itrs // len(train_loader) + 1
    for epoch in range(max_epoch):
        for img, label_true in train_loader:
            cur_itrs += 1
            run([img, label_true], cur_itrs)
            _check_peers(args, cur_itrs)
            if _due(cur_itrs, args, test_loader):
                best(model, st.optimizer, st.lr_scheduler, test_loader, cur_itrs)
            if cur_itrs > args.total_itrs:
                _check_peers(args, cur_itrs, final=True)
                return slog.losses()
    _check_peers(args, cur_itrs, final=True)
    return slog.losses()


def Mean_Teacher(model, ema_model, label_loader, unlabel_loader, test_loader, args):
    """2017_03_NIPS_Mean-Teacher_ACDC.py:63-162."""
    st = MeanTeacherStep(model, ema_model, args, getattr(args, "dp", None))
    best, best_ema = _Best(args, "model", "model_save_path"), _Best(args, "ema", "ema_model_save_path")
    model.train()
    ema_model.train()          # the teacher stays in train mode (2017_03...py:70)
    w = [0.0]
    emit = lambda row, h: {"mean_teacher/loss": row[0], "mean_teacher/lr": h["lr"], "mean_teacher/consistency_weight": h["w"]}
    slog = ScalarLog(torch.device(args.device), 8, emit, _writer(args), _log_every(args))
    run = _LoopRunner(st, args, slog, ["parts"], lambda: {"lr": st._lr(st.optimizer), "w": float(st.sc.host[S_COEF_A + 4])})
    cur_itrs = 0
    labels = _cycle(label_loader)
    max_epoch = args.total_itrs // len(unlabel_loader) + 1
    for epoch in range(max_epoch):
        for unlabel_img, _ in unlabel_loader:
            cur_itrs += 1
            label_img, target_label = next(labels)
            run([label_img, target_label, unlabel_img], cur_itrs)
            _check_peers(args, cur_itrs)
            if _due(cur_itrs, args, test_loader):
                best(model, st.optimizer, st.lr_scheduler, test_loader, cur_itrs, "model")
                best_ema(ema_model, st.optimizer, st.lr_scheduler, test_loader, cur_itrs, "ema")
            if cur_itrs > args.total_itrs:
                _check_peers(args, cur_itrs, final=True)
                return slog.losses()
    _check_peers(args, cur_itrs, final=True)
    return slog.losses()


def _teacher_student_loop(st, tag, model, ema_model, label_loader, unlabel_loader, test_loader, args, extra_inputs, extra_scalars=None):
    """The loop the teacher / student drivers besides Mean_Teacher share (2022_02_ISBI_ICT-MedSeg_ACDC.py:93-190,
    2019_07_MICCAI_Uncertainty_Aware_ACDC.py:109-217): one unlabelled batch per iteration, the labelled loader cycled beside it, the
    per-iteration random inputs drawn on the host side of the (captured) step in the reference's order, both networks evaluated every
    ``step_size`` iterations.  Scalars: <tag>/loss, /lr, /consistency_weight, /consistency_loss [+ extra_scalars(st)]."""
    best, best_ema = _Best(args, "model", "model_save_path"), _Best(args, "ema", "ema_model_save_path")
    model.train()
    ema_model.train()

    def emit(row, h):
        d = {f"{tag}/loss": row[0], f"{tag}/lr": h["lr"], f"{tag}/consistency_weight": h["w"], f"{tag}/consistency_loss": row[5]}
        d.update({f"{tag}/{k}": v for k, v in h.items() if k not in ("lr", "w")})
        return d

    def host():
        h = {"lr": st._lr(st.optimizer), "w": float(st.sc.host[S_COEF_A + 4])}
        if extra_scalars is not None:
            h.update(extra_scalars(st))
        return h

    slog = ScalarLog(torch.device(args.device), 8, emit, _writer(args), _log_every(args))
    run = _LoopRunner(st, args, slog, ["parts"], host)
    cur_itrs = 0
    labels = _cycle(label_loader)
    max_epoch = args.total_itrs // len(unlabel_loader) + 1
    for epoch in range(max_epoch):
        for unlabel_img, _ in unlabel_loader:
            cur_itrs += 1
            label_img, target_label = next(labels)
            run([label_img, target_label, unlabel_img] + list(extra_inputs(unlabel_img)), cur_itrs)
            _check_peers(args, cur_itrs)
            if _due(cur_itrs, args, test_loader):
                best(model, st.optimizer, st.lr_scheduler, test_loader, cur_itrs, "test_model")
                best_ema(ema_model, st.optimizer, st.lr_scheduler, test_loader, cur_itrs, "test_ema_model")
            if cur_itrs > args.total_itrs:
                _check_peers(args, cur_itrs, final=True)
                return slog.losses()
    _check_peers(args, cur_itrs, final=True)
    return slog.losses()


def ICT_MedSeg(model, ema_model, label_loader, unlabel_loader, test_loader, args):
    """2022_02_ISBI_ICT-MedSeg_ACDC.py:65-190: the mix factors are numpy Beta draws per iteration (:112) -- an input of the step."""
    st = ICTStep(model, ema_model, args, getattr(args, "dp", None))
    return _teacher_student_loop(st, "ICT_MedSeg", model, ema_model, label_loader, unlabel_loader, test_loader, args,
                                 lambda u: [st.draw_mix_factors(u.shape[0])])


def Uncertainty_Aware(model, ema_model, label_loader, unlabel_loader, test_loader, args):
    """2019_07_MICCAI_Uncertainty_Aware_ACDC.py:82-217: the noise fields of the teacher passes are device draws per iteration (:130, :142),
    in the reference's order -- inputs of the step (the T / 2 doubled-batch fields travel stacked)."""
    st = UAMTStep(model, ema_model, args, getattr(args, "dp", None))

    def noise(u):
        n0, rest = st.draw_noise(u.to(args.device).float())
        return [n0, torch.stack(rest)]

    return _teacher_student_loop(st, "Uncertainty_Aware", model, ema_model, label_loader, unlabel_loader, test_loader, args, noise,
                                 lambda s: {"threshold": float(s.sc.host[S_THRESH])})


def CPS(model1, model2, label_loader, unlabel_loader, test_loader, args, step_cls=None):
    """2021_06_CVPR_CPS_ACDC.py:61-169 (and, with CTCTStep, 2021_12_MIDL_CTCT_ACDC.py:68-214)."""
    st = (step_cls or CPSStep)(model1, model2, args, getattr(args, "dp", None))
    best1, best2 = _Best(args, "model1", "model1_save_path"), _Best(args, "model2", "model2_save_path")
    model1.train()
    model2.train()
    ctct = isinstance(st, CTCTStep)

    def emit(row, h):          # rows: [parts1 (8) | parts2 (8)]
        loss = row[0] + row[8]
        if ctct:               # 2021_12...py:158-161
            return {"mynet/loss": loss, "mynet/lr1": h["lr1"], "mynet/lr2": h["lr2"], "mynet/consistency_weight": h["w"]}
        sup = _sup_of(row, 0) + _sup_of(row, 8)          # 2021_06...py:124-128
        return {"mynet/loss": loss, "mynet/lr": h["lr1"], "mynet/consistency_weight": h["w"], "mynet/loss_semi": loss - sup, "mynet/loss_sup": sup}

    slog = ScalarLog(torch.device(args.device), 16, emit, _writer(args), _log_every(args))
    wpos = S_COEF_A + 3
    run = _LoopRunner(st, args, slog, ["parts1", "parts2"],
                      lambda: {"lr1": st._lr(st.optimizer1), "lr2": st._lr(st.optimizer2),
                               "w": float(st.sc.host[wpos]) * (1.0 if ctct else 2.0)})          # CPS parks 0.5 * w there (0.5 * w * (CE + Dice))
    cur_itrs = 0
    labels = _cycle(label_loader)
    max_epoch = args.total_itrs // len(unlabel_loader) + 1
    for epoch in range(max_epoch):
        for unlabel_img, _ in unlabel_loader:
            cur_itrs += 1
            label_img, target_label = next(labels)
            run([label_img, target_label, unlabel_img], cur_itrs)
            _check_peers(args, cur_itrs)
            if _due(cur_itrs, args, test_loader):
                best1(model1, st.optimizer1, st.lr_scheduler1, test_loader, cur_itrs, "model1")
                best2(model2, st.optimizer2, st.lr_scheduler2, test_loader, cur_itrs, "model2")
            if cur_itrs > args.total_itrs:
                _check_peers(args, cur_itrs, final=True)
                return slog.losses()
    _check_peers(args, cur_itrs, final=True)
    return slog.losses()


def CTCT(model1, model2, label_loader, unlabel_loader, test_loader, args):
    """Cross teaching between CNN and transformer with the driver's signature (2021_12_MIDL_CTCT_ACDC.py:82)."""
    return CPS(model1, model2, label_loader, unlabel_loader, test_loader, args, step_cls=CTCTStep)


def S4CVnet(model1, model2, ema_model, label_loader, unlabel_loader, test_loader, args):
    """2022_08_CVPR_S4CVNet_ACDC.py:70-230."""
    st = S4CVNetStep(model1, model2, ema_model, args, getattr(args, "dp", None))
    best1, best2 = _Best(args, "model1", "model1_save_path"), _Best(args, "model2", "model2_save_path")
    best_ema = _Best(args, "ema", "ema_model_save_path")
    model1.train()
    model2.train()

    def emit(row, h):          # 2022_08_CVPR_S4CVNet_ACDC.py:174-180
        loss, sup = row[0] + row[8], _sup_of(row, 0) + _sup_of(row, 8)
        return {"S4CVnet/loss": loss, "S4CVnet/loss_semi": loss - sup, "S4CVnet/loss_sup": sup, "S4CVnet/lr1": h["lr1"], "S4CVnet/lr2": h["lr2"],
                "S4CVnet/consistency_weight_cps": h["w"], "S4CVnet/consistency_weight_mt": h["w"]}

    slog = ScalarLog(torch.device(args.device), 16, emit, _writer(args), _log_every(args))
    run = _LoopRunner(st, args, slog, ["parts1", "parts2"],
                      lambda: {"lr1": st._lr(st.optimizer1), "lr2": st._lr(st.optimizer2), "w": float(st.sc.host[S_COEF_A + 3]) / 7.0})
    cur_itrs = 0
    labels = _cycle(label_loader)
    max_epoch = args.total_itrs // len(unlabel_loader) + 1
    for epoch in range(max_epoch):
        for img_unlabel, _ in unlabel_loader:
            cur_itrs += 1
            img_labeled, target_label = next(labels)
            noise = st.draw_noise(img_unlabel.to(args.device).float())      # an INPUT of the (captured) step: drawn here, in the reference's order (:109)
            run([img_labeled, target_label, img_unlabel, noise], cur_itrs)
            _check_peers(args, cur_itrs)
            if _due(cur_itrs, args, test_loader):
                best1(model1, st.optimizer1, st.lr_scheduler1, test_loader, cur_itrs, "model1")
                best2(model2, st.optimizer2, st.lr_scheduler2, test_loader, cur_itrs, "model2")
                best_ema(ema_model, st.optimizer2, st.lr_scheduler2, test_loader, cur_itrs, "ema")
            if cur_itrs > args.total_itrs:
                _check_peers(args, cur_itrs, final=True)
                return slog.losses()
    _check_peers(args, cur_itrs, final=True)
    return slog.losses()


def HPFG(model1, model2, ema_model, label_loader, unlabel_loader, test_loader, args):
    """main.py:79-289."""
    st = HPFGStep(model1, model2, ema_model, args, getattr(args, "dp", None))
    best1, best2 = _Best(args, "model1", "model1_save_path"), _Best(args, "model2", "model2_save_path")
    best_ema = _Best(args, "ema", "ema_model_save_path")
    model1.train()
    model2.train()

    def emit(row, h):          # rows: [parts1 (8) | parts2 (8) | contrast]; main.py:216-222
        loss = row[0] + row[8] + h["w"] * row[16]
        sup = _sup_of(row, 0) + _sup_of(row, 8)
        return {"HPFG/loss": loss, "HPFG/loss_semi": loss - sup, "HPFG/loss_sup": sup, "HPFG/lr1": h["lr1"], "HPFG/lr2": h["lr2"],
                "HPFG/consistency_weight_cps": h["w"], "HPFG/consistency_weight_mt": h["w"]}

    slog = ScalarLog(torch.device(args.device), 17, emit, _writer(args), _log_every(args))
    run = _LoopRunner(st, args, slog, ["parts1", "parts2", "contrast"],
                      lambda: {"lr1": st._lr(st.optimizer1), "lr2": st._lr(st.optimizer2), "w": float(st._w)})
    cur_itrs = 0
    it_a, it_b = _cycle(label_loader), _cycle(label_loader)      # two independent labelled iterators (main.py:119-120)
    max_epoch = args.total_itrs // len(unlabel_loader) + 1
    for epoch in range(max_epoch):
        for img_unlabel, _ in unlabel_loader:
            cur_itrs += 1
            label_img, target_label = next(it_a)
            label_img1, target_label1 = next(it_b)
            nl, nu = label_img.shape[0], img_unlabel.shape[0]
            rep = nu // nl
            cm = st.make_cutmix_mask(nu, (args.train_crop_size[0], args.train_crop_size[1]), device=torch.device(args.device))
            run([label_img, target_label, label_img1.repeat(rep, 1, 1, 1), target_label1.repeat(rep, 1, 1), img_unlabel, cm], cur_itrs)
            _check_peers(args, cur_itrs)
            if _due(cur_itrs, args, test_loader):
                best1(model1, st.optimizer1, st.lr_scheduler1, test_loader, cur_itrs, "model1")
                best2(model2, st.optimizer2, st.lr_scheduler2, test_loader, cur_itrs, "model2")
                best_ema(ema_model, st.optimizer2, st.lr_scheduler2, test_loader, cur_itrs, "model1")      # main.py:259-272 saves optimizer2 with it
            if cur_itrs > args.total_itrs:
                _check_peers(args, cur_itrs, final=True)
                return slog.losses()
    _check_peers(args, cur_itrs, final=True)
    return slog.losses()
