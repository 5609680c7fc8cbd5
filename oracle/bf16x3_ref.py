"""Oracle (test infrastructure): CPU emulation of the HIP path's split-bf16 ("bf16x3") convolution arithmetic.

The gfx950 kernels of hpfg_amd/csrc evaluate every product of a convolution (forward, input gradient, weight gradient) as
hi*hi + hi*lo + lo*hi with x = hi + lo, hi = bf16(x) (round to nearest even), lo = bf16(x - hi), and accumulate in fp32
(DESIGN.md section 5).  This module restates exactly that error model on the CPU -- the same three partial products, each an exact
fp32 convolution of bf16-valued operands -- so that a test can measure what the arithmetic itself does to a trajectory
(emulated oracle vs nominal fp32 oracle: the committed CONTROL of tests/test_gpu_steps.py) separately from what a kernel does
(HIP result vs emulated oracle: accumulation order only).

Which convolutions are split follows the kernels: every 3x3 / 1x1 conv of the U-Net except the FORWARD of the first layer
(conv_first_mfma_kernel multiplies in exact fp32; its weight gradient goes through the split-bf16 fused backward kernel).
The projection necks and the losses are exact fp32 on both sides.

The reference (/root/reference) has no counterpart of this file: it is the arithmetic model of the MI355X build, used only as a checker.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F
from torch.nn import grad as G

_MODE = "f32"


def mode() -> str:
    return _MODE


@contextlib.contextmanager
def math_mode(m: str):
    """with math_mode("bf16x3"): every oracle conv (oracle.unet_ref) uses the split-bf16 emulation."""
    global _MODE
    assert m in ("f32", "bf16x3", "f64acc", "bf16x3_f64acc")
    prev, _MODE = _MODE, m
    try:
        yield
    finally:
        _MODE = prev


def split(t: torch.Tensor):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def _conv3(a_hi, a_lo, b_hi, b_lo, op, acc64=False):
    """op(a, b) for the three kept partial products (the lo*lo term, 2^-16 relative, is dropped as on the device).
    acc64: the same products accumulated in fp64 and rounded once (the "f64acc" controls: what is left between this and the fp32
    accumulation is summation-order noise, the only thing that separates a correct kernel from the oracle)."""
    if acc64:
        a_hi, a_lo, b_hi, b_lo = a_hi.double(), a_lo.double(), b_hi.double(), b_lo.double()
        return (op(a_hi, b_hi) + op(a_hi, b_lo) + op(a_lo, b_hi)).float()
    return op(a_hi, b_hi) + op(a_hi, b_lo) + op(a_lo, b_hi)


class _ConvSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, padding: int, split_fwd: bool, acc64: bool):
        ctx.save_for_backward(x, w)
        ctx.padding, ctx.has_bias, ctx.acc64 = padding, b is not None, acc64
        if split_fwd:
            xh, xl = split(x)
            wh, wl = split(w)
            y = _conv3(xh, xl, wh, wl, lambda a, c: F.conv2d(a, c, None, padding=padding), acc64)
        elif acc64:
            y = F.conv2d(x.double(), w.double(), None, padding=padding).float()
        else:
            y = F.conv2d(x, w, None, padding=padding)
        return y if b is None else y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        p = ctx.padding
        dyh, dyl = split(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wh, wl = split(w)
            dx = _conv3(dyh, dyl, wh, wl, lambda g, c: G.conv2d_input(x.shape, c, g, padding=p), ctx.acc64)
        if ctx.needs_input_grad[1]:
            xh, xl = split(x)
            dw = _conv3(xh, xl, dyh, dyl, lambda a, g: G.conv2d_weight(a, w.shape, g, padding=p), ctx.acc64)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None, None


def conv2d(x, w, b=None, padding: int = 0, first_layer: bool = False):
    """F.conv2d(x, w, b, padding=padding) in the current math mode (see module docstring)."""
    if _MODE == "f32":
        return F.conv2d(x, w, b, padding=padding)
    if _MODE == "f64acc":          # exact products, fp64 accumulation, one rounding: the fp32 oracle up to summation-order noise
        return F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=padding).float()
    return _ConvSplit.apply(x, w, b, padding, not first_layer, _MODE == "bf16x3_f64acc")
