"""Optimizer / scheduler factories with the reference's signatures (utils/__init__.py:13-49).

``build_optimizer(args, model)`` returns ``FusedSGD`` for an hpfg_amd U-Net on the GPU: torch.optim.SGD's law
(momentum, weight decay on every parameter, no dampening / nesterov) as ONE HIP kernel over the model's flat parameter,
gradient and momentum buffers, with the learning rate read from device memory so the step can live inside a hipGraph.
It is a ``torch.optim.Optimizer`` (param_groups, state_dict, zero_grad), so the reference's LR schedulers drive it unchanged.
"""
from __future__ import annotations

import torch

from .. import _lib as L
from .scheduler import CosineWarmupLR_Scheduler, Medical_LR, PolyLR


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, model, lr=0.01, momentum=0.0, weight_decay=0.0):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        flat = model.flat_params
        self._mom = torch.zeros_like(flat)
        self._lr_host = torch.zeros(1, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.zeros(1)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.grad_scale = 1.0
        # torch.optim.SGD skips a parameter whose .grad is None (no weight decay, no momentum).  A step that never back-propagates into a
        # trailing part of the flat buffer (HPFG's first student: its projection necks' outputs are discarded, main.py:152) says so here.
        self.active_numel = None

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_flat_grad()          # one memset; p.grad views stay attached
        self.model.attach_grad_views()

    def push_lr(self):
        """Stage the current lr on the device (async copy from pinned memory; capturable)."""
        self._lr_host[0] = float(self.param_groups[0]["lr"])
        self._lr_dev.copy_(self._lr_host, non_blocking=True)

    @torch.no_grad()
    def step(self, closure=None, push_lr: bool = True, ema=None):
        """ema: optional (teacher_flat, numel, alpha_dev) -- the EMA teacher update of the same iteration rides along in the same launch
        (hpfg_sgd_ema_step; bit-identical to step() followed by utils.update_ema_variables)."""
        g = self.param_groups[0]
        flat, grad = self.model.flat_params, self.model.flat_grads
        if self._mom.data_ptr() == 0 or self._mom.numel() != flat.numel() or self._mom.device != flat.device:
            self._mom = torch.zeros_like(flat)
        if push_lr:
            self.push_lr()
        st = torch.cuda.current_stream(flat.device).cuda_stream
        n_sgd = flat.numel() if self.active_numel is None else int(self.active_numel)
        if ema is not None:
            t, n_ema, alpha_dev = ema
            assert t.is_cuda and t.dtype == torch.float32 and t.numel() == flat.numel() and 0 <= n_ema <= n_sgd
            L.check(L.load().hpfg_sgd_ema_step(L.ptr(flat), L.ptr(grad), L.ptr(self._mom), n_sgd, L.ptr(self._lr_dev), float(g["momentum"]),
                                               float(g["weight_decay"]), float(self.grad_scale), L.ptr(t), int(n_ema), L.ptr(alpha_dev), st),
                    "sgd_ema_step")
            return
        L.check(L.load().hpfg_sgd_step(L.ptr(flat), L.ptr(grad), L.ptr(self._mom), n_sgd, L.ptr(self._lr_dev), float(g["momentum"]),
                                       float(g["weight_decay"]), float(self.grad_scale), st), "sgd_step")

    def state_dict(self):
        d = super().state_dict()
        d["flat_momentum"] = self._mom
        return d

    def load_state_dict(self, sd):
        sd = dict(sd)
        mom = sd.pop("flat_momentum", None)
        super().load_state_dict(sd)
        if mom is not None:
            self._mom.copy_(mom)


def flatten_parameters(model):
    """Re-point every parameter of `model` at a slice of ONE flat fp32 buffer (gradients are packed into a second flat buffer per step), so
    that an optimizer step is a single kernel.  The hpfg_amd U-Nets are built that way; this does it for any module (SegFormer).
    Must run after the module is on its device.  Sets model.flat_params / model.flat_grads."""
    if hasattr(model, "flat_params"):
        return
    ps = [p for p in model.parameters()]
    dev = ps[0].device
    n = sum(p.numel() for p in ps)
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    grad = torch.zeros(n, dtype=torch.float32, device=dev)
    o = 0
    with torch.no_grad():
        for p in ps:
            k = p.numel()
            flat[o:o + k].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + k].view(p.shape)
            o += k
    model.flat_params, model.flat_grads = flat, grad
    model._hpfg_generic_flat = True      # gradients reach flat_grads only through FusedAdamW.gather_flat_grads()


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(lr, weight_decay) -- betas (0.9, 0.999), eps 1e-8 as the reference leaves them (utils/__init__.py:17-19) -- as ONE
    HIP kernel over the model's flat parameter / gradient / moment buffers; learning rate and step count live on the device, so the
    update replays inside a hipGraph.  A ``torch.optim.Optimizer`` (param_groups, zero_grad), so the LR schedulers drive it unchanged."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8):
        flatten_parameters(model)
        self.model = model
        super().__init__([p for p in model.parameters() if p.requires_grad], dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        flat = model.flat_params
        self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        self._step_dev = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._lr_host = torch.zeros(1, dtype=torch.float32).pin_memory()
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.grad_scale = 1.0
        # WHO packs the per-parameter gradients into model.flat_grads is a property of the step object, fixed when it is built -- not a
        # per-call flag: a captured step evaluates host flags once, at capture time (round-4 advisor finding: graph_b captured step() with
        # the flag False, so every replay re-concatenated the UN-reduced p.grad over the all-reduced buffer).  external_gather = True
        # (data parallel: _StepBase._own_gather): the gradient exchange gathers, reduces in place, and step() consumes flat_grads as they
        # are; False (one rank): step() gathers.
        self.external_gather = False
        model._hpfg_flat_optimizer = self

    def zero_grad(self, set_to_none: bool = True):
        # grads are dropped, not zeroed: autograd then ASSIGNS each parameter's gradient (accumulating into kept views would cost one
        # add kernel per parameter tensor, 189 launches); gather_flat_grads() packs them into the flat buffer in one concat
        for p in self.model.parameters():
            p.grad = None

    @torch.no_grad()
    def gather_flat_grads(self):
        ps = [p for p in self.model.parameters()]
        torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps], out=self.model.flat_grads)
        return self.model.flat_grads

    def push_lr(self):
        self._lr_host[0] = float(self.param_groups[0]["lr"])
        self._lr_dev.copy_(self._lr_host, non_blocking=True)

    @torch.no_grad()
    def step(self, closure=None, push_lr: bool = True):
        g = self.param_groups[0]
        flat = self.model.flat_params
        grad = self.model.flat_grads if self.external_gather else self.gather_flat_grads()
        if push_lr:
            self.push_lr()
        st = torch.cuda.current_stream(flat.device).cuda_stream
        L.check(L.load().hpfg_adamw_step(L.ptr(flat), L.ptr(grad), L.ptr(self._m), L.ptr(self._v), flat.numel(), L.ptr(self._lr_dev), L.ptr(self._step_dev),
                                         float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(self.grad_scale), st),
                "adamw_step")

    def state_dict(self):
        """torch's dict plus the flat Adam moments and the device step count (resuming must not reset them; the reference saves
        ``optimizer.state_dict()`` in its checkpoints, main.py:262-272)."""
        d = super().state_dict()
        d["flat_m"], d["flat_v"], d["step"] = self._m, self._v, self._step_dev
        return d

    def load_state_dict(self, sd):
        sd = dict(sd)
        m, v, st = sd.pop("flat_m", None), sd.pop("flat_v", None), sd.pop("step", None)
        super().load_state_dict(sd)
        if m is not None:
            self._m.copy_(m)
        if v is not None:
            self._v.copy_(v)
        if st is not None:
            self._step_dev.copy_(torch.as_tensor(st, dtype=torch.float32).reshape(1))


def build_optimizer(args, model):
    if args.opt == "sgd":
        if hasattr(model, "flat_params") and model.flat_params.is_cuda:
            return FusedSGD(model, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
        return torch.optim.SGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if args.opt == "adamW":
        params = list(model.parameters())
        if params and params[0].is_cuda:
            return FusedAdamW(model, lr=args.lr, weight_decay=args.weight_decay)
        return torch.optim.AdamW(params, lr=args.lr, weight_decay=args.weight_decay)
    if args.opt == "adam":
        return torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    raise ValueError("get_optimizer error")


def build_lr_scheduler(args, optimizer):
    if args.sched == "cosine":
        return CosineWarmupLR_Scheduler(optimizer=optimizer, base_lr=args.lr, warmup_epochs=args.warmup_epochs, warmup_lr=args.warmup_lr,
                                        final_lr=args.min_lr, iter_per_epoch=args.step_size, num_epochs=args.total_itrs // args.step_size)
    if args.sched == "poly":
        return PolyLR(optimizer, max_iters=args.total_itrs, power=0.1, min_lr=args.min_lr)
    if args.sched == "medical":
        return Medical_LR(optimizer=optimizer, base_lr=args.lr, max_iterations=args.total_itrs)
    raise ValueError("get_lr_scheduler error")
