"""Token-layout kernels of the SegFormer branch (csrc/tokens.hip) against plain torch fp32 on the CPU: values and gradients."""
import pytest
import torch
import torch.nn.functional as F

from hpfg_amd.ops_tokens import attention, bn_relu_dropout, dwconv_gelu, im2col, layer_norm, linear, residual_scale, resize_bilinear
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("rows,C_", [((2, 49), 32), ((3, 100), 160), ((1, 7), 256), ((2, 5, 5), 64), ((1, 3), 1024)])
def test_layer_norm(rows, C_):
    g = torch.Generator().manual_seed(C_)
    x = torch.randn(*rows, C_, generator=g) * 2 + 0.5
    w, b = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    dy = torch.randn(*rows, C_, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (C_,), wr, br).backward(dy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = layer_norm(xd, wd, bd)
    y.backward(dy.to(DEV))
    assert maxerr(y.detach().cpu(), F.layer_norm(x, (C_,), w, b)) < 2e-5
    assert maxerr(xd.grad.cpu(), xr.grad) < 5e-5
    assert maxerr(wd.grad.cpu(), wr.grad) < 2e-4 and maxerr(bd.grad.cpu(), br.grad) < 2e-4


@pytest.mark.parametrize("math", ["bf16x3", "f32"])
@pytest.mark.parametrize("B,N,M,heads", [(2, 256, 4, 1), (2, 64, 4, 2), (1, 16, 4, 5), (2, 4, 4, 8), (1, 300, 49, 2), (1, 49, 49, 8), (1, 10, 64, 1),
                                         (2, 3136, 49, 1), (1, 784, 49, 2), (3, 1100, 33, 5)])
def test_attention_core(B, N, M, heads, math):
    """softmax(scale q k^T) v and its three gradients: the MFMA kernels (csrc/attn.hip, split-bf16 products: default math mode) and the exact-fp32
    thread-per-query kernels (HPFG_MATH=f32) against plain PyTorch fp32."""
    from hpfg_amd import ops_tokens
    g = torch.Generator().manual_seed(N + M)
    C_ = heads * 32
    q, kv, do = torch.randn(B, N, C_, generator=g), torch.randn(B, M, 2 * C_, generator=g), torch.randn(B, N, C_, generator=g)
    scale = 32 ** -0.5

    def ref(q_, kv_):
        qh = q_.reshape(B, N, heads, 32).permute(0, 2, 1, 3)
        k, v = kv_.reshape(B, M, 2, heads, 32).permute(2, 0, 3, 1, 4)
        a = ((qh @ k.transpose(-2, -1)) * scale).softmax(-1)
        return (a @ v).transpose(1, 2).reshape(B, N, C_)
    qr, kr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    ref(qr, kr).backward(do)
    qd, kd = q.to(DEV).requires_grad_(True), kv.to(DEV).requires_grad_(True)
    ops_tokens.MATH["mode"] = math
    try:
        out = attention(qd, kd, heads, scale)
        out.backward(do.to(DEV))
    finally:
        ops_tokens.MATH["mode"] = None
    k = 1.0 if math == "f32" else 8.0          # split-bf16 products: 2^-17 relative per product instead of fp32 rounding
    assert maxerr(out.detach().cpu(), ref(q, kv)) < 2e-5 * k
    assert maxerr(qd.grad.cpu(), qr.grad) < 5e-5 * k
    assert maxerr(kd.grad.cpu(), kr.grad) < 2e-4 * k * max(1.0, (N / 256) ** 0.5)        # dK / dV sum over all N queries


@pytest.mark.parametrize("B,H,W,C_", [(2, 16, 16, 128), (2, 8, 8, 256), (1, 4, 4, 640), (2, 2, 2, 1024), (1, 7, 5, 128)])
def test_dwconv_gelu(B, H, W, C_):
    g = torch.Generator().manual_seed(C_ + H)
    x = torch.randn(B, H, W, C_, generator=g)
    w, b = torch.randn(C_, 1, 3, 3, generator=g) * 0.3, torch.randn(C_, generator=g) * 0.1
    dy = torch.randn(B, H, W, C_, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.gelu(F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1, groups=C_)).permute(0, 2, 3, 1)
    yr.backward(dy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = dwconv_gelu(xd, wd, bd)
    y.backward(dy.to(DEV))
    assert maxerr(y.detach().cpu(), yr.detach()) < 2e-5
    assert maxerr(xd.grad.cpu(), xr.grad) < 5e-5
    assert maxerr(wd.grad.cpu(), wr.grad) < 5e-4 and maxerr(bd.grad.cpu(), br.grad) < 5e-4


@pytest.mark.parametrize("B,h,w,H,W,C_", [(2, 8, 8, 16, 16, 256), (2, 4, 4, 16, 16, 256), (1, 2, 2, 16, 16, 256), (2, 14, 14, 56, 56, 4), (1, 7, 7, 56, 56, 8),
                                          (1, 5, 3, 20, 12, 4)])
def test_resize_bilinear(B, h, w, H, W, C_):
    g = torch.Generator().manual_seed(h * 100 + H)
    x = torch.randn(B, h, w, C_, generator=g)
    dy = torch.randn(B, H, W, C_, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    yr.backward(dy)
    xd = x.to(DEV).requires_grad_(True)
    y = resize_bilinear(xd, H, W)
    y.backward(dy.to(DEV))
    assert maxerr(y.detach().cpu(), yr.detach()) < 1e-5
    assert maxerr(xd.grad.cpu(), xr.grad) < 2e-5


@pytest.mark.parametrize("B,H,W,C_,lows", [(2, 56, 56, 256, [(28, 28), (14, 14), (7, 7)]), (3, 24, 20, 8, [(12, 10)]), (1, 16, 16, 64, [(8, 8), (4, 4)])])
def test_resize_sum(B, H, W, C_, lows):
    """base + sum_k interpolate(x_k) in one pass (the head's fused map, model/segformer.py:309-314 with linear_fuse applied per stage)."""
    from hpfg_amd.ops_tokens import resize_sum
    g = torch.Generator().manual_seed(H * 7 + len(lows))
    base = torch.randn(B, H, W, C_, generator=g)
    xs = [torch.randn(B, h, w, C_, generator=g) for h, w in lows]
    dy = torch.randn(B, H, W, C_, generator=g)
    br, xr = base.clone().requires_grad_(True), [x.clone().requires_grad_(True) for x in xs]
    yr = br
    for x in xr:
        yr = yr + F.interpolate(x.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    yr.backward(dy)
    bd, xd = base.to(DEV).requires_grad_(True), [x.to(DEV).requires_grad_(True) for x in xs]
    y = resize_sum(bd, *xd)
    y.backward(dy.to(DEV))
    assert maxerr(y.detach().cpu(), yr.detach()) < 2e-5
    assert maxerr(bd.grad.cpu(), br.grad) == 0.0
    for a_, r_ in zip(xd, xr):
        assert maxerr(a_.grad.cpu(), r_.grad) < 5e-5


@pytest.mark.parametrize("B,N,C_,use_mask", [(2, 256, 256, True), (3, 49, 256, False), (2, 1000, 64, True)])
def test_bn_relu_dropout(B, N, C_, use_mask):
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, C_, generator=g) * 1.5 + 0.3
    w, b = torch.randn(C_, generator=g), torch.randn(C_, generator=g) * 0.3
    dy = torch.randn(B, N, C_, generator=g)
    mask = torch.empty(B, C_, 1, 1).bernoulli_(0.9, generator=g) if use_mask else None
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    mu = xr.mean((0, 1))
    var = (xr - mu).square().mean((0, 1))
    yr = torch.relu((xr - mu) * torch.rsqrt(var + 1e-5) * wr + br)
    if use_mask:
        yr = yr * mask.reshape(B, 1, C_) / 0.9
    yr.backward(dy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y, m_, v_ = bn_relu_dropout(xd, wd, bd, None if mask is None else mask.to(DEV), 0.9)
    y.backward(dy.to(DEV))
    assert maxerr(y.detach().cpu(), yr.detach()) < 2e-5
    assert maxerr(m_.cpu(), mu.detach()) < 1e-5 and maxerr(v_.cpu(), var.detach()) < 1e-4
    assert maxerr(xd.grad.cpu(), xr.grad) < 5e-5
    assert maxerr(wd.grad.cpu(), wr.grad) < 1e-3 and maxerr(bd.grad.cpu(), br.grad) < 1e-3


@pytest.mark.parametrize("R,N,K", [(9000, 32, 32), (20000, 64, 32), (8200, 32, 128), (10000, 256, 64), (8192, 4, 256), (9001, 160, 36)])
def test_linear_tall_weight_gradient(R, N, K):
    g = torch.Generator().manual_seed(N + K)
    x, w, b = torch.randn(2, R // 2, K, generator=g), torch.randn(N, K, generator=g) * 0.2, torch.randn(N, generator=g)
    x = x[:, : R // 2]
    dy = torch.randn(2, R // 2, N, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.linear(xr, wr, br).backward(dy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = linear(xd, wd, bd)
    y.backward(dy.to(DEV))
    scale = float(wr.grad.abs().max())
    assert maxerr(y.detach().cpu(), F.linear(x, w, b)) < 1e-4
    assert maxerr(xd.grad.cpu(), xr.grad) < 1e-4
    assert maxerr(wd.grad.cpu(), wr.grad) < 2e-5 * max(1.0, scale) * 10 and maxerr(bd.grad.cpu(), br.grad) < 1e-3


@pytest.mark.parametrize("B,H,W,C_,k,s", [(2, 64, 64, 1, 7, 4), (2, 16, 16, 32, 3, 2), (1, 9, 7, 64, 3, 2), (1, 224, 224, 1, 7, 4), (2, 8, 8, 160, 3, 2)])
def test_im2col_matches_conv(B, H, W, C_, k, s):
    """im2col + GEMM == nn.Conv2d(k, s, padding k//2), values and input gradient."""
    g = torch.Generator().manual_seed(H + C_)
    x = torch.randn(B, H, W, C_, generator=g)
    w, b = torch.randn(8, C_, k, k, generator=g) * 0.2, torch.randn(8, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), w, b, stride=s, padding=k // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd = x.to(DEV).requires_grad_(True)
    y = F.linear(im2col(xd, k, s), w.permute(0, 2, 3, 1).reshape(8, -1).to(DEV), b.to(DEV))          # [B, L, 8]
    y.backward(dy.flatten(2).transpose(1, 2).to(DEV))
    assert maxerr(y.detach().cpu(), yr.detach().flatten(2).transpose(1, 2)) < 1e-4
    assert maxerr(xd.grad.cpu(), xr.grad) < 1e-4


def test_residual_scale():
    g = torch.Generator().manual_seed(1)
    x, y, s = torch.randn(3, 50, 32, generator=g), torch.randn(3, 50, 32, generator=g), torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9]).view(3, 1, 1)
    d = torch.randn(3, 50, 32, generator=g)
    for sc in (s, None):
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        (xr + (yr * sc if sc is not None else yr)).backward(d)
        xd, yd = x.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
        out = residual_scale(xd, yd, None if sc is None else sc.to(DEV))
        out.backward(d.to(DEV))
        assert maxerr(out.detach().cpu(), x + (y * sc if sc is not None else y)) < 1e-6
        assert maxerr(xd.grad.cpu(), xr.grad) < 1e-6 and maxerr(yd.grad.cpu(), yr.grad) < 1e-6


@pytest.mark.parametrize("R,N,K", [(1568, 32, 2048), (1568, 256, 1024), (6272, 640, 160), (25088, 128, 64), (392, 160, 640), (100, 4, 256), (3136, 32, 49), (130, 36, 20)])
@pytest.mark.parametrize("math", ["bf16x3", "f32"])
def test_linear_on_the_hip_gemms(R, N, K, math):
    """nn.Linear / 1x1 conv of the SegFormer branch on hpfg_gemm_bf16x3 (split-bf16 MFMA; all three products: X W^T + b, dY W, dY^T X)
    and, where an operand is not 4-aligned (the 7x7x1 patch embedding: K = 49) or HPFG_MATH=f32, on the exact-fp32 MFMA GEMM.  The 1568-token
    shapes have few output tiles and a long contraction: they run K-split (hpfg_gemm_bf16x3_splitk, 16 / 8 / 2 splits)."""
    from hpfg_amd import ops_tokens
    g = torch.Generator().manual_seed(R + N)
    x, w, b = torch.randn(R, K, generator=g), torch.randn(N, K, generator=g) * 0.2, torch.randn(N, generator=g)
    dy = torch.randn(R, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    yr.backward(dy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    ops_tokens.MATH["mode"] = math
    try:
        y = linear(xd, wd, bd)
        y.backward(dy.to(DEV))
    finally:
        ops_tokens.MATH["mode"] = None
    rel = 3e-5 if math == "bf16x3" else 2e-6              # per-product 2^-17 (split-bf16) / fp32 rounding, accumulated over K or R terms
    assert maxerr(y.detach().cpu().double(), yr.detach()) < rel * K ** 0.5 * 3
    assert maxerr(xd.grad.cpu().double(), xr.grad) < rel * N ** 0.5 * 3
    assert maxerr(wd.grad.cpu().double(), wr.grad) < rel * R ** 0.5 * 4 * 3
    assert maxerr(bd.grad.cpu().double(), br.grad) < 2e-6 * R ** 0.5 * 4 * 3
