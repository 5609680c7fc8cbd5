"""GPU: conv_thin_kernel (whole-tile staging forward conv of the thin 16-pixel-aligned layers) against conv_bf16x3_kernel on the same
virtual inputs -- the raw output must be bit-identical (same staged values, same MFMA order; the concat loader's bilinear blend to 1 ulp), the BatchNorm partial sums equal up to the
order of the per-workgroup rows -- and against plain PyTorch fp32 on the CPU."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hpfg_amd import _lib as L
from tests.helpers import AdHocConv, maxerr, nchw, stream
from tests.test_gpu_kernels import _bn_table, _materialize
from tests.test_gpu_fused_bwd import _bnact

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
OLD = 0x4000      # any flag bit in `math` keeps a launch on conv_bf16x3_kernel


def _conv(layer, a0, a1, N, H, W, math, stats):
    lib = L.load()
    ca = L.ConvArgs()
    ca.a0, ca.a1, ca.math = a0, (a1 if a1 is not None else L.Act()), math
    ca.wpk, ca.bias = L.ptr(layer.wpk16_f), L.ptr(layer.bias_pad)
    out = torch.full((N, H, W, layer.cout), float("nan"), device=DEV)
    ca.out, ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = L.ptr(out), layer.cout, layer.cout, layer.cout_pad, N, H, W, 9
    part = None
    if stats:
        part = torch.zeros(lib.hpfg_conv_stat_blocks(N, H, W) * 2 * layer.cout_pad, device=DEV)
        ca.stat_partials = L.ptr(part)
        rows = lib.hpfg_conv_stat_rows(C.byref(ca))
        assert 0 < rows * 2 * layer.cout_pad <= part.numel()
    L.check(lib.hpfg_conv_fwd(C.byref(ca), stream(DEV)), "conv_fwd")
    torch.cuda.synchronize()
    if stats:
        part = part[: rows * 2 * layer.cout_pad].view(rows, 2, layer.cout_pad).double().sum(0).cpu()
    return out, part


CASES = [  # N, H, W, cin, cout, input kind, dropout p
    (2, 32, 48, 16, 16, "bnact", 0.05), (3, 32, 32, 16, 16, "bnact", 0.0), (2, 32, 32, 16, 4, "bnact", 0.0),
    (2, 32, 16, 16, 32, "pool", 0.0), (2, 32, 32, 32, 16, "cat", 0.0), (9, 64, 64, 16, 16, "bnact", 0.0), (5, 48, 64, 32, 16, "cat", 0.0),
]


@pytest.mark.parametrize("N,H,W,cin,cout,ak,p", CASES)
def test_thin_conv_equals_the_chunked_kernel_bitwise(N, H, W, cin, cout, ak, p):
    g = torch.Generator().manual_seed(H * 5 + cin + cout)
    layer = AdHocConv(cin, cout, 9, DEV, seed=cin * 3 + cout, hw=(H, W))
    a1 = None
    if ak == "bnact":
        z = torch.randn(N, H, W, cin, generator=g).to(DEV)
        tab = _bn_table(cin, 3).to(DEV)
        a0 = _bnact(z, tab, cin, H, W, p=p, seed=55)
    elif ak == "pool":
        z = torch.randn(N, 2 * H, 2 * W, cin, generator=g).to(DEV)
        tab = _bn_table(cin, 3).to(DEV)
        a0 = _bnact(z, tab, cin, 2 * H, 2 * W, mode=L.ACT_BNACT_POOL)
    else:
        c2 = cin // 2
        z = torch.randn(N, H, W, c2, generator=g).to(DEV)
        tab = _bn_table(c2, 3).to(DEV)
        a0 = _bnact(z, tab, c2, H, W)
        ud = torch.randn(N, H // 2, W // 2, c2, generator=g).to(DEV)
        a1 = L.Act()
        a1.z, a1.mode, a1.C, a1.Hs, a1.Ws, a1.pstride = L.ptr(ud), L.ACT_UP2X, c2, H // 2, W // 2, c2
    stats = cout % 16 == 0
    out, part = _conv(layer, a0, a1, N, H, W, L.MATH_BF16X3, stats)
    ref, rpart = _conv(layer, a0, a1, N, H, W, L.MATH_BF16X3 | OLD, stats)
    if ak == "cat":      # the bilinear blend of the upsampled half is contracted into FMAs differently by the two compilations: 1 ulp on inputs
        assert maxerr(out.cpu(), ref.cpu()) < 2e-5 * max(1.0, float(ref.abs().max()))
    else:
        assert torch.equal(out, ref), f"raw output differs: {maxerr(out.cpu(), ref.cpu())}"
    if stats:
        assert maxerr(part, rpart) < 1e-4 * max(1.0, float(rpart.abs().max()))
    a_in = _materialize(a0, a1, N, H, W, cin)
    want = F.conv2d(nchw(a_in), layer.w.cpu(), layer.b.cpu(), padding=1)
    assert maxerr(nchw(out.cpu()), want) < 3e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("N,H,W,p", [(2, 32, 48, 0.0), (5, 48, 32, 0.2)])
def test_thin_dgrad_of_the_64_to_32_layer_equals_the_chunked_kernel(N, H, W, p):
    """dgrad of a 64 -> 32 channel layer (decoder.up3's first conv at 112^2): dZ source of 32 channels, 64 output channels, weights in LDS."""
    from tests.test_gpu_fused_bwd import _dz
    g = torch.Generator().manual_seed(H + W)
    layer = AdHocConv(64, 32, 9, DEV, seed=4, hw=(H, W))
    zo = torch.randn(N, H, W, 32, generator=g).to(DEV)
    dA = torch.randn(N, H, W, 32, generator=g).to(DEV)
    tabo = _bn_table(32, 11).to(DEV)
    d = _dz(zo, tabo, dA, 32, H, W, p=p, seed=21)
    def dgrad(math):
        ca = L.ConvArgs()
        ca.a0, ca.a1, ca.math, ca.wpk = d, L.Act(), math, L.ptr(layer.wpk16_d)
        out = torch.full((N, H, W, 64), float("nan"), device=DEV)
        ca.out, ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = L.ptr(out), 64, 64, 64, N, H, W, 9
        L.check(L.load().hpfg_conv_fwd(C.byref(ca), stream(DEV)), "dgrad")
        torch.cuda.synchronize()
        return out

    got, ref = dgrad(L.MATH_BF16X3), dgrad(L.MATH_BF16X3 | OLD)
    assert torch.equal(got, ref), f"dgrad differs: {maxerr(got.cpu(), ref.cpu())}"
    dz = _materialize(d, None, N, H, W, 32)
    xr = torch.zeros(N, 64, H, W, requires_grad=True)
    F.conv2d(xr, layer.w.cpu(), None, padding=1).backward(nchw(dz))
    assert maxerr(nchw(got.cpu()), xr.grad) < 3e-4 * max(1.0, float(xr.grad.abs().max()))
