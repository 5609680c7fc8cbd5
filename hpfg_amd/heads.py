"""Projection necks of UNet_Plus (reference model/unet.py:120-152) and the NT-Xent core of Dense_Loss (utils/loss/dense_loss.py:17-40)
on the HIP library.

GAP -> Linear -> ReLU -> Linear and AdaptiveAvgPool(4x4) -> 1x1 conv -> ReLU -> 1x1 conv on the [N,256,14,14] bottleneck and on the
[N,4,224,224] logits: both poolings are ONE kernel (``hpfg_neck_pool_fwd``), every dense product -- forward, input gradient, weight
gradient -- is ``hpfg_gemm_f32`` (exact fp32 on the matrix cores: the necks are ~1 GFLOP per HPFG step), bias gradients are column
sums.  The gradient w.r.t. the neck inputs flows back into the HIP U-Net backward (engine.backward's ``dfeat4`` / ``dlogits``).
No rocBLAS / MIOpen call is left on the HPFG step.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _st(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def gemm(a: torch.Tensor, sam: int, sak: int, b: torch.Tensor, sbk: int, sbn: int, m: int, n: int, k: int, bias=None, relu: bool = False,
         out: torch.Tensor = None) -> torch.Tensor:
    """out[m,n] = act(sum_k a(m,k) b(k,n) + bias[n]) with explicit element strides (see include/hpfg_hip.h: hpfg_gemm_f32)."""
    if not a.is_cuda:
        raise RuntimeError("hpfg_amd necks run on the HIP library only (no CPU fallback)")
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    lib = L.load()
    splits = lib.hpfg_gemm_f32_splits(m, n, k)          # few output tiles, long contraction (the 2048-wide layers): split K over workgroups
    scratch = torch.empty(splits * m * n, dtype=torch.float32, device=a.device) if splits > 1 else None
    L.check(lib.hpfg_gemm_f32_splitk(L.ptr(a), sam, sak, L.ptr(b), sbk, sbn, L.ptr(out), n, m, n, k, L.ptr(bias), 1 if relu else 0, 0, L.ptr(scratch), _st(a)),
            "gemm_f32")
    return out


class _Linear(torch.autograd.Function):
    """y = relu?(x W^T + b); x [R,K] contiguous, W [M,K] contiguous (an nn.Linear weight or a flattened 1x1 conv weight)."""

    @staticmethod
    def forward(ctx, x, w, b, relu: bool, wp=None, bp=None):
        """wp / bp (optional): the Parameters behind w / b of a network in ``direct_grads`` mode (one zero_grad + one backward per step): backward
        then WRITES dW / db into their .grad views of the flat gradient buffer -- no temporary, no AccumulateGrad add per tensor (16 small
        launches per U-Net+ and step on the chain of backward) -- and hands autograd nothing for them."""
        x, w = x.contiguous(), w.contiguous()
        R, K = x.shape
        M = w.shape[0]
        y = gemm(x, K, 1, w, 1, K, R, M, K, bias=b, relu=relu)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu = relu
        ctx.direct = (wp, bp) if (wp is not None and bp is not None and wp.grad is not None and bp.grad is not None) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        R, K = x.shape
        M = w.shape[0]
        lib = L.load()
        dy = dy.contiguous()
        if ctx.relu:
            if ctx.direct is None:
                dy = dy.clone()      # (direct mode: dy is the next layer's freshly made dx and has no other reader -- masked in place)
            L.check(lib.hpfg_relu_bwd(L.ptr(dy), L.ptr(y), dy.numel(), _st(dy)), "relu_bwd")
        dx = gemm(dy, M, 1, w, K, 1, R, K, M) if ctx.needs_input_grad[0] else None              # dY W
        scratch = torch.empty(lib.hpfg_col_sum_splits(R) * M, dtype=torch.float32, device=dy.device)
        if ctx.direct is not None:
            wp, bp = ctx.direct
            gemm(dy, 1, M, x, K, 1, M, K, R, out=wp.grad.view(M, K))                             # dY^T X, straight into the flat gradient buffer
            L.check(lib.hpfg_col_sum2(L.ptr(dy), R, M, M, L.ptr(bp.grad), L.ptr(scratch), _st(dy)), "col_sum2")
            return dx, None, None, None, None, None
        dw = gemm(dy, 1, M, x, K, 1, M, K, R)                                                    # dY^T X
        db = torch.empty(M, dtype=torch.float32, device=dy.device)
        L.check(lib.hpfg_col_sum2(L.ptr(dy), R, M, M, L.ptr(db), L.ptr(scratch), _st(dy)), "col_sum2")
        return dx, dw, db, None, None, None


class _NeckPool(torch.autograd.Function):
    """x [N,C,H,W] (NHWC memory preferred) -> (gap [N,C], pool [N*S*S, C]) = AdaptiveAvgPool2d(1) and (S) (unet.py:141-142,146)."""

    @staticmethod
    def forward(ctx, x, s: int):
        v = x.permute(0, 2, 3, 1)
        if not v.is_contiguous():
            v = v.contiguous()
        N, H, W, C = v.shape
        gap = torch.empty(N, C, dtype=torch.float32, device=x.device)
        pool = torch.empty(N * s * s, C, dtype=torch.float32, device=x.device)
        L.check(L.load().hpfg_neck_pool_fwd(L.ptr(v), C, N, H, W, C, s, L.ptr(gap), L.ptr(pool), _st(x)), "neck_pool_fwd")
        ctx.shape, ctx.s = (N, H, W, C), s
        return gap, pool

    @staticmethod
    def backward(ctx, dgap, dpool):
        N, H, W, C = ctx.shape
        dx = torch.empty(N, H, W, C, dtype=torch.float32, device=(dgap if dgap is not None else dpool).device)
        dg = dgap.contiguous() if dgap is not None else None
        dp = dpool.contiguous() if dpool is not None else None
        L.check(L.load().hpfg_neck_pool_bwd(L.ptr(dg), L.ptr(dp), N, H, W, C, ctx.s, L.ptr(dx), _st(dx)), "neck_pool_bwd")
        return dx.permute(0, 3, 1, 2), None


def projection_neck(m, x: torch.Tensor, s: int = 4, direct: bool = False):
    """x: [N,C,H,W] (any strides).  Returns (g [N,128], d [N,128,s*s]) like projection_conv.forward (unet.py:139-152); d is a view of
    [N, s*s, 128] memory (position-major), which Dense_Loss consumes without a copy.
    direct: the network is in ``direct_grads`` mode and its neck parameters' .grad are views of the flat gradient buffer: see _Linear."""
    if not x.is_cuda:
        raise RuntimeError("hpfg_amd necks run on the HIP library only (no CPU fallback)")
    gap, pool = _NeckPool.apply(x.float(), s)

    def lin(x_, layer, relu, flat=False):
        w, b = layer.weight, layer.bias
        return _Linear.apply(x_, w.flatten(1) if flat else w, b, relu, w if direct else None, b if direct else None)

    g = lin(lin(gap, m.mlp["0"], True), m.mlp["2"], False)
    d = lin(lin(pool, m.mlp_conv["0"], True, True), m.mlp_conv["2"], False, True)
    N = x.shape[0]
    return g, d.view(N, s * s, d.shape[1]).permute(0, 2, 1)


class _NTXent(torch.autograd.Function):
    """contrastive_loss(out_1 = a (student, differentiable), out_2 = b (teacher, constant)) of dense_loss.py:17-36."""

    @staticmethod
    def forward(ctx, a, b, temperature: float):
        if not a.is_cuda:
            raise RuntimeError("hpfg_amd.Dense_Loss runs on the HIP library only (no CPU fallback)")
        lib = L.load()
        a, b = a.float(), b.float()
        if a.dim() == 2:
            a3, b3 = a.unsqueeze(2), b.unsqueeze(2)
        else:
            a3, b3 = a.flatten(2), b.flatten(2)
        n, D, S = a3.shape
        # position-major memory ([N, S, D], what projection_neck produces) is used as it is; anything else in the canonical [N, D, S] order
        pm = S > 1 and a3.permute(0, 2, 1).is_contiguous() and b3.permute(0, 2, 1).is_contiguous()
        if not pm:
            a3, b3 = a3.contiguous(), b3.contiguous()
        sd, ss = (1, D) if pm else (S, 1)
        F_ = D * S
        U = torch.empty(2 * n, F_, dtype=torch.float32, device=a.device)
        norms = torch.empty(2 * n * S, dtype=torch.float32, device=a.device)
        st = _st(a)
        L.check(lib.hpfg_l2norm_fwd(L.ptr(a3), n, D, S, sd, ss, L.ptr(U), L.ptr(norms), st), "l2norm_fwd")
        L.check(lib.hpfg_l2norm_fwd(L.ptr(b3), n, D, S, sd, ss, U.data_ptr() + 4 * n * F_, norms.data_ptr() + 4 * n * S, st), "l2norm_fwd")
        gram = gemm(U, F_, 1, U, 1, F_, 2 * n, 2 * n, F_)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        Q = torch.empty(n, 2 * n, dtype=torch.float32, device=a.device)
        L.check(lib.hpfg_ntxent_rows(L.ptr(gram), n, float(temperature), L.ptr(loss), L.ptr(Q), st), "ntxent_rows")
        ctx.save_for_backward(U, Q, norms)
        ctx.geo = (n, D, S, sd, ss, pm, tuple(a.shape))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        U, Q, norms = ctx.saved_tensors
        n, D, S, sd, ss, pm, shape = ctx.geo
        F_ = D * S
        dU = gemm(Q, 2 * n, 1, U, F_, 1, n, F_, 2 * n)                                    # Q U: gradient w.r.t. the normalised student rows
        dx = torch.empty(n, F_, dtype=torch.float32, device=U.device)
        g = gout.reshape(1).float().contiguous()
        L.check(L.load().hpfg_l2norm_bwd(L.ptr(dU), L.ptr(U), L.ptr(norms), n, D, S, sd, ss, L.ptr(g), L.ptr(dx), _st(U)), "l2norm_bwd")
        dx = dx.view(n, S, D).permute(0, 2, 1) if pm else dx.view(n, D, S)
        return dx.reshape(shape), None, None


class _GatherRows(torch.autograd.Function):
    """all-gather of feature rows over the data-parallel group (rank-major); backward hands back this rank's rows of the gradient.
    Every rank then evaluates the SAME global-batch loss, and the SUM all-reduce of the parameter gradients counts each row once."""

    @staticmethod
    def forward(ctx, x, dp):
        x = x.contiguous()
        ctx.rank, ctx.n = dp.rank, x.shape[0]
        if getattr(dp, "p2p_grads", False) and x.is_cuda and x.dtype == torch.float32 and dp.world_size * x.numel() <= dp.grad_floats:
            # through the peer windows (kernels on this stream, capturable -- the HPFG step's global-batch mode then is ONE hipGraph): every rank
            # puts its rows into its own slot of a zero buffer and the SUM all-reduce of csrc/peer.hip yields the concatenation (x + 0 is exact)
            buf = torch.zeros((dp.world_size,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            buf[dp.rank].copy_(x)
            dp.peer_allreduce_sum(buf.view(-1))
            return buf.view((dp.world_size * x.shape[0],) + tuple(x.shape[1:]))
        import torch.distributed as dist
        parts = [torch.empty_like(x) for _ in range(dp.world_size)]
        dist.all_gather(parts, x, group=dp.group)
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None


def ntxent(a: torch.Tensor, b: torch.Tensor, temperature: float, dp=None) -> torch.Tensor:
    """dp (a DataParallelContext in the global-batch mode, sync_bn=True, with more than one rank): the contrast runs over the features of the
    WHOLE global batch, as Dense_Loss(batch_size + unlabel_batch_size) does in the single-process reference (main.py:89,172) -- two small
    all-gathers ([N/R, 128] and [N/R, 2048] rows) per call; without it (per-rank mode) each rank contrasts its own batch."""
    b = b.detach()
    if dp is not None and getattr(dp, "sync_bn", True) and dp.world_size > 1:
        a, b = _GatherRows.apply(a, dp), _GatherRows.apply(b, dp)
    return _NTXent.apply(a, b, temperature)
