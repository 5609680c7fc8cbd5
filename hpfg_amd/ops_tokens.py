"""Autograd bindings of the token-layout HIP kernels (csrc/tokens.hip) used by the SegFormer branch: LayerNorm, the attention core
softmax(q k^T) v, and depthwise-3x3 + GELU.  Device tensors only; the library raises if it is missing (no fallback)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib as L


def _st(t):
    return torch.cuda.current_stream(t.device).cuda_stream


MATH = {"mode": None}


def gemm_math() -> str:
    """Arithmetic of the token GEMMs and the attention core: HPFG_MATH (default bf16x3), like the U-Net convolutions."""
    import os
    return MATH["mode"] or os.environ.get("HPFG_MATH", "bf16x3")


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"hpfg_amd.{what} runs on the HIP library only (no CPU fallback)")


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        _need_gpu(x, "layer_norm")
        lib = L.load()
        xc = x.contiguous().float()
        C_ = xc.shape[-1]
        rows = xc.numel() // C_
        y = torch.empty_like(xc)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L.check(lib.hpfg_ln_fwd(L.ptr(xc), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd), rows, C_, _st(x)), "ln_fwd")
        ctx.save_for_backward(xc, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, gamma, mean, rstd = ctx.saved_tensors
        C_ = x.shape[-1]
        rows = x.numel() // C_
        dyc = dy.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty(2, C_, dtype=torch.float32, device=x.device)          # [dgamma ; dbeta]: one reduction launch fills both
        dg, db = dgb[0], dgb[1]
        part = torch.empty(lib.hpfg_ln_bwd_blocks(rows) * 2 * C_, dtype=torch.float32, device=x.device)
        L.check(lib.hpfg_ln_bwd(L.ptr(x), L.ptr(dyc), L.ptr(gamma), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(part), rows, C_,
                                _st(x)), "ln_bwd")
        return dx, dg, db


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm(C) with eps 1e-5 over the last dimension."""
    return _LayerNorm.apply(x, gamma, beta)


class _Attention(torch.autograd.Function):
    """softmax(scale q k^T) v on the matrix cores (csrc/attn.hip): split-bf16 MFMA products in the default math mode; HPFG_MATH=f32 keeps
    the exact-fp32 one-thread-per-query kernels of csrc/tokens.hip (dK / dV then through the exact GEMM)."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        _need_gpu(q, "attention")
        lib = L.load()
        qc, kvc = q.contiguous().float(), kv.contiguous().float()
        B, N, C_ = qc.shape
        M = kvc.shape[1]
        out = torch.empty_like(qc)
        ctx.math = gemm_math()
        if ctx.math == "bf16x3":
            L.check(lib.hpfg_attn_mfma_fwd(L.ptr(qc), L.ptr(kvc), L.ptr(out), B, N, M, heads, scale, _st(q)), "attn_mfma_fwd")
        else:
            L.check(lib.hpfg_attn_fwd(L.ptr(qc), L.ptr(kvc), L.ptr(out), B, N, M, heads, scale, _st(q)), "attn_fwd")
        ctx.save_for_backward(qc, kvc)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        q, kv = ctx.saved_tensors
        B, N, C_ = q.shape
        M, h = kv.shape[1], ctx.heads
        d = C_ // h
        do = dout.contiguous()
        dq = torch.empty_like(q)
        if ctx.math == "bf16x3":
            dkv = torch.empty_like(kv)
            scratch = torch.empty(B * h * lib.hpfg_attn_mfma_blocks(N) * 2 * 64 * 32, dtype=torch.float32, device=q.device)
            L.check(lib.hpfg_attn_mfma_bwd(L.ptr(q), L.ptr(kv), L.ptr(do), L.ptr(dq), L.ptr(dkv), L.ptr(scratch), B, N, M, h, ctx.scale, _st(q)),
                    "attn_mfma_bwd")
            return dq, dkv, None, None
        P = torch.empty(B, h, N, M, dtype=torch.float32, device=q.device)
        dS = torch.empty_like(P)
        L.check(lib.hpfg_attn_bwd(L.ptr(q), L.ptr(kv), L.ptr(do), L.ptr(dq), L.ptr(P), L.ptr(dS), B, N, M, h, ctx.scale, _st(q)), "attn_bwd")
        # dV = P^T dO, dK = scale * dS^T Q per (image, head): exact-fp32 MFMA GEMMs (hpfg_gemm_f32) writing straight into the [B,M,2,h,d] layout
        dkv = torch.empty(B, M, 2, h, d, dtype=torch.float32, device=q.device)
        st = _st(q)
        for b in range(B):
            for hh in range(h):
                pbh, sbh = P[b, hh], dS[b, hh]                      # [N, M] each: A(m, k) = X[k, m] -> sam = 1, sak = M
                L.check(lib.hpfg_gemm_f32(L.ptr(pbh), 1, M, do.data_ptr() + 4 * (b * N * C_ + hh * d), C_, 1, L.ptr(dkv[b, 0, 1, hh]), 2 * C_,
                                          M, d, N, None, 0, 0, st), "attn dV")
                L.check(lib.hpfg_gemm_f32(L.ptr(sbh), 1, M, q.data_ptr() + 4 * (b * N * C_ + hh * d), C_, 1, L.ptr(dkv[b, 0, 0, hh]), 2 * C_,
                                          M, d, N, None, 0, 0, st), "attn dK")
        dkv[:, :, 0].mul_(ctx.scale)
        dkv = dkv.view(B, M, 2 * C_)
        return dq, dkv, None, None


def attention(q: torch.Tensor, kv: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """softmax(scale * q k^T) v per head.  q [B,N,C], kv [B,M,2C] laid out [.., 2, heads, C/heads] (the kv Linear's output); C/heads == 32."""
    assert q.shape[-1] // heads == 32 and kv.shape[-1] == 2 * q.shape[-1] and kv.shape[1] <= 64
    return _Attention.apply(q, kv, heads, scale)


class _DWGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, "dwconv_gelu")
        lib = L.load()
        xc = x.contiguous().float()
        B, H, W, C_ = xc.shape
        w9 = weight.reshape(C_, 9).t().contiguous()
        y = torch.empty_like(xc)
        L.check(lib.hpfg_dwgelu_fwd(L.ptr(xc), L.ptr(w9), L.ptr(bias), L.ptr(y), B, H, W, C_, _st(x)), "dwgelu_fwd")
        ctx.save_for_backward(xc, w9, bias)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, w9, bias = ctx.saved_tensors
        B, H, W, C_ = x.shape
        dyc = dy.contiguous()
        du, dx = torch.empty_like(x), torch.empty_like(x)
        dwb = torch.empty(10, C_, dtype=torch.float32, device=x.device)         # [dw9 (9 rows) ; dbias]: one reduction launch fills both
        dw9, db = dwb[:9], dwb[9]
        part = torch.empty(lib.hpfg_dwgelu_bwd_blocks(B, H, W) * 10 * C_, dtype=torch.float32, device=x.device)
        L.check(lib.hpfg_dwgelu_bwd(L.ptr(x), L.ptr(w9), L.ptr(bias), L.ptr(dyc), L.ptr(du), L.ptr(dx), L.ptr(dw9), L.ptr(db), L.ptr(part), B, H, W, C_,
                                    _st(x)), "dwgelu_bwd")
        return dx, dw9.t().reshape(ctx.wshape), db


def dwconv_gelu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """gelu(depthwise_conv3x3(x) + bias) on x [B,H,W,C] (NHWC); weight [C,1,3,3] as nn.Conv2d(C, C, 3, 1, 1, groups=C) holds it."""
    return _DWGelu.apply(x, weight, bias)


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        _need_gpu(x, "resize_bilinear")
        xc = x.contiguous().float()
        B, h, w, C_ = xc.shape
        y = torch.empty(B, H, W, C_, dtype=torch.float32, device=x.device)
        L.check(L.load().hpfg_resize_bilinear_fwd(L.ptr(xc), L.ptr(y), B, h, w, H, W, C_, _st(x)), "resize_fwd")
        ctx.shape = (B, h, w, H, W, C_)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, h, w, H, W, C_ = ctx.shape
        dyc = dy.contiguous()
        dx = torch.empty(B, h, w, C_, dtype=torch.float32, device=dy.device)
        L.check(L.load().hpfg_resize_bilinear_bwd(L.ptr(dyc), L.ptr(dx), B, h, w, H, W, C_, _st(dy)), "resize_bwd")
        return dx, None, None


class _ResizeSum(torch.autograd.Function):
    """base [B,H,W,C] + sum_k resize(x_k [B,h_k,w_k,C]) in one pass (hpfg_resize_sum_fwd); backward: the gradient itself for base, the
    resize backward of it for every x_k."""

    @staticmethod
    def forward(ctx, base, *xs):
        _need_gpu(base, "resize_sum")
        bc = base.contiguous().float()
        B, H, W, C_ = bc.shape
        xc = [x.contiguous().float() for x in xs]
        assert 1 <= len(xc) <= 3 and all(x.shape[0] == B and x.shape[3] == C_ for x in xc)
        y = torch.empty_like(bc)
        ptrs = (C.c_void_p * len(xc))(*[x.data_ptr() for x in xc])
        hs, ws = (C.c_int * len(xc))(*[x.shape[1] for x in xc]), (C.c_int * len(xc))(*[x.shape[2] for x in xc])
        L.check(L.load().hpfg_resize_sum_fwd(L.ptr(bc), ptrs, hs, ws, len(xc), L.ptr(y), B, H, W, C_, _st(base)), "resize_sum_fwd")
        ctx.geo = (B, H, W, C_, [(x.shape[1], x.shape[2]) for x in xc])
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C_, lows = ctx.geo
        dyc = dy.contiguous()
        outs = [dyc if ctx.needs_input_grad[0] else None]
        for k, (h, w) in enumerate(lows):
            if not ctx.needs_input_grad[1 + k]:
                outs.append(None)
                continue
            dx = torch.empty(B, h, w, C_, dtype=torch.float32, device=dy.device)
            L.check(L.load().hpfg_resize_bilinear_bwd(L.ptr(dyc), L.ptr(dx), B, h, w, H, W, C_, _st(dy)), "resize_bwd")
            outs.append(dx)
        return tuple(outs)


def resize_sum(base: torch.Tensor, *xs: torch.Tensor) -> torch.Tensor:
    """base + sum_k F.interpolate(x_k, size=base's, mode="bilinear", align_corners=False), NHWC, C % 4 == 0, up to 3 addends."""
    return _ResizeSum.apply(base, *xs)


def resize_bilinear(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """F.interpolate(..., size=(H, W), mode="bilinear", align_corners=False) on NHWC x [B,h,w,C] (C % 4 == 0), upsampling."""
    return _Resize.apply(x, H, W)


class _BNReLUDrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mask, keep, eps):
        _need_gpu(x, "bn_relu_dropout")
        lib = L.load()
        B, N, C_ = x.shape
        xc = x.contiguous().float()
        R = B * N
        part = torch.empty(lib.hpfg_tok_stat_blocks(R) * 2 * C_, dtype=torch.float32, device=x.device)
        sums = torch.empty(2, C_, dtype=torch.float32, device=x.device)
        L.check(lib.hpfg_tok_col_stats(L.ptr(xc), R, C_, L.ptr(part), L.ptr(sums), _st(x)), "tok_col_stats")
        mean = sums[0] / R
        var = (sums[1] / R - mean * mean).clamp_min_(0.0)
        rstd = torch.rsqrt(var + eps)
        y = torch.empty_like(xc)
        m = None if mask is None else mask.reshape(B, C_).float().contiguous()
        L.check(lib.hpfg_bnrelu_apply(L.ptr(xc), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), L.ptr(m), 1.0 / keep, N, L.ptr(y), R, C_, _st(x)),
                "bnrelu_apply")
        ctx.save_for_backward(xc, gamma, beta, mean, rstd, m if m is not None else xc.new_empty(0))
        ctx.keep, ctx.N = keep, N
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        lib = L.load()
        x, gamma, beta, mean, rstd, m = ctx.saved_tensors
        B, N, C_ = x.shape
        R = B * N
        dyc = dy.contiguous()
        dx = torch.empty_like(x)
        part = torch.empty(lib.hpfg_tok_stat_blocks(R) * 2 * C_, dtype=torch.float32, device=x.device)
        sums = torch.empty(2, C_, dtype=torch.float32, device=x.device)
        L.check(lib.hpfg_bnrelu_bwd(L.ptr(x), L.ptr(dyc), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), L.ptr(m if m.numel() else None),
                                    1.0 / ctx.keep, N, L.ptr(dx), L.ptr(part), L.ptr(sums), R, C_, _st(x)), "bnrelu_bwd")
        return dx, sums[1].clone(), sums[0].clone(), None, None, None


def bn_relu_dropout(x, gamma, beta, mask=None, keep: float = 0.9, eps: float = 1e-5):
    """Train-mode BatchNorm over all tokens of x [B,N,C] + ReLU + per-(image, channel) dropout mask [B,C] (0/1, scaled by 1/keep).
    Returns (y, batch mean, biased batch variance); the caller updates the running statistics."""
    return _BNReLUDrop.apply(x, gamma, beta, mask, keep, eps)


TALL_ROWS = 8192          # from this many tokens on, the weight gradient of a Linear runs on the row-split HIP kernel


def _gemm(a, sam, sak, b, sbk, sbn, m, n, k, bias=None, math="bf16x3"):
    """out[m,n] = sum_k a(m,k) b(k,n) + bias[n] on the HIP library: split-bf16 MFMA when the operands allow it (unit stride along one
    index, 4-element alignment) and `math` asks for it, exact-fp32 MFMA otherwise (include/hpfg_hip.h: hpfg_gemm_bf16x3 / hpfg_gemm_f32)."""
    lib = L.load()
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    if math == "bf16x3" and lib.hpfg_gemm_bf16x3_ok(L.ptr(a), sam, sak, L.ptr(b), sbk, sbn, m, n, k):
        splits = lib.hpfg_gemm_bf16x3_splits(m, n, k)      # few output tiles + long K (the 1568-token layers): split the contraction
        if splits > 1:
            scratch = torch.empty(splits * m * n, dtype=torch.float32, device=a.device)
            L.check(lib.hpfg_gemm_bf16x3_splitk(L.ptr(a), sam, sak, L.ptr(b), sbk, sbn, L.ptr(out), n, m, n, k, L.ptr(bias), 0, 0, L.ptr(scratch), _st(a)),
                    "gemm_bf16x3_splitk")
        else:
            L.check(lib.hpfg_gemm_bf16x3(L.ptr(a), sam, sak, L.ptr(b), sbk, sbn, L.ptr(out), n, m, n, k, L.ptr(bias), 0, 0, _st(a)), "gemm_bf16x3")
    else:
        L.check(lib.hpfg_gemm_f32(L.ptr(a), sam, sak, L.ptr(b), sbk, sbn, L.ptr(out), n, m, n, k, L.ptr(bias), 0, 0, _st(a)), "gemm_f32")
    return out


class _Linear(torch.autograd.Function):
    """y = x W^T + b over tokens x [.., K] -- nn.Linear, the kernel == stride spatial-reduction conv, the patch embeddings after im2col and
    the head's 1x1 convs of reference model/segformer.py -- entirely on the HIP library: forward and dX on hpfg_gemm_bf16x3 (or the exact
    fp32 GEMM), dW on the row-split deterministic kernel when there are many tokens (tall-skinny dY^T X) and on the GEMM otherwise, db as
    a column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, "linear")
        N, K = weight.shape
        xc = x.reshape(-1, K).contiguous().float()
        w = weight.contiguous()
        math = gemm_math()
        y = _gemm(xc, K, 1, w, 1, K, xc.shape[0], N, K, bias, math)
        ctx.save_for_backward(xc, w)
        ctx.has_bias, ctx.math, ctx.xshape = bias is not None, math, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        xc, w = ctx.saved_tensors
        N, K = w.shape
        R = xc.shape[0]
        dyc = dy.reshape(-1, N).contiguous()
        dx = _gemm(dyc, N, 1, w, K, 1, R, K, N, None, ctx.math).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        db = None
        if ctx.math == "bf16x3":          # dY^T X on MFMA; the bias gradient (column sums of dY) comes out of the same pass
            wb = torch.empty(N * K + (N if ctx.has_bias else 0), dtype=torch.float32, device=w.device)
            part = torch.empty(lib.hpfg_gemm_tn_splits(R, N, K) * wb.numel(), dtype=torch.float32, device=w.device)
            L.check(lib.hpfg_gemm_tn_bf16x3(L.ptr(dyc), L.ptr(xc), L.ptr(wb), L.ptr(part), R, N, K, 1 if ctx.has_bias else 0, _st(w)), "gemm_tn_bf16x3")
            dw = wb[:N * K].view(N, K)
            if ctx.has_bias:
                db = wb[N * K:]
            return dx, dw, db
        if R >= TALL_ROWS and N % 4 == 0 and K % 4 == 0:
            dw = torch.empty_like(w)
            part = torch.empty(lib.hpfg_linear_wgrad_splits(R, N, K) * N * K, dtype=torch.float32, device=w.device)
            L.check(lib.hpfg_linear_wgrad(L.ptr(dyc), L.ptr(xc), L.ptr(dw), L.ptr(part), R, N, K, _st(w)), "linear_wgrad")
        else:
            dw = _gemm(dyc, 1, N, xc, K, 1, N, K, R, None, ctx.math)                   # dY^T X, exact fp32
        if ctx.has_bias:
            db = torch.empty(N, dtype=torch.float32, device=w.device)
            scratch = torch.empty(lib.hpfg_col_sum_splits(R) * N, dtype=torch.float32, device=w.device)
            L.check(lib.hpfg_col_sum2(L.ptr(dyc), R, N, N, L.ptr(db), L.ptr(scratch), _st(w)), "col_sum2")
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """F.linear on the HIP library (no rocBLAS call): see _Linear."""
    return _Linear.apply(x, weight, bias)


class _Im2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s):
        _need_gpu(x, "im2col")
        xc = x.contiguous().float()
        B, H, W, C_ = xc.shape
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        cols = torch.empty(B, Ho * Wo, k * k * C_, dtype=torch.float32, device=x.device)
        L.check(L.load().hpfg_im2col_nhwc(L.ptr(xc), L.ptr(cols), B, H, W, C_, k, s, _st(x)), "im2col")
        ctx.geo = (B, H, W, C_, k, s)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        B, H, W, C_, k, s = ctx.geo
        dc = dcols.contiguous()
        dx = torch.empty(B, H, W, C_, dtype=torch.float32, device=dc.device)
        L.check(L.load().hpfg_col2im_nhwc(L.ptr(dc), L.ptr(dx), B, H, W, C_, k, s, _st(dc)), "col2im")
        return dx, None, None


def im2col(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """Patches of a conv with kernel k, stride s, padding k // 2 over NHWC x [B,H,W,C] -> [B, Ho*Wo, k*k*C], patch order (u, v, c)."""
    return _Im2col.apply(x, k, s)


class _ResidualScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, scale):
        _need_gpu(x, "residual_scale")
        xc, yc = x.contiguous().float(), y.contiguous().float()
        B = xc.shape[0]
        per = xc.numel() // B
        out = torch.empty_like(xc)
        sc = None if scale is None else scale.reshape(B).float().contiguous()
        L.check(L.load().hpfg_residual_scale(L.ptr(xc), L.ptr(yc), L.ptr(sc), L.ptr(out), B, per, _st(x)), "residual_scale")
        ctx.sc, ctx.geo = sc, (B, per)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.sc is None:
            return dout, dout, None
        B, per = ctx.geo
        d = dout.contiguous()
        dy = torch.empty_like(d)
        L.check(L.load().hpfg_scale_rows(L.ptr(d), L.ptr(ctx.sc), L.ptr(dy), B, per, _st(d)), "scale_rows")
        return dout, dy, None


def residual_scale(x: torch.Tensor, y: torch.Tensor, scale=None) -> torch.Tensor:
    """x + y * scale[b] per sample b (scale None = 1): a residual branch with stochastic depth in one kernel."""
    return _ResidualScale.apply(x, y, scale)
