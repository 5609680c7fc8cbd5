"""GPU: hpfg_fused_bwd (one kernel: dgrad + BatchNorm-backward sums + weight-gradient slabs of a thin 3x3 layer) against the separate
hpfg_conv_fwd (dgrad) and hpfg_wgrad launches on the same virtual sources, and against plain PyTorch fp32 autograd on the CPU."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hpfg_amd import _lib as L
from tests.helpers import AdHocConv, maxerr, nchw, plain_act, stream
from tests.test_gpu_kernels import _bn_table, _materialize

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _bnact(z, tab, C_, H, W, mode=L.ACT_BNACT, p=0.0, seed=0):
    a = L.Act()
    a.z, a.bn, a.mode, a.C, a.Hs, a.Ws, a.pstride, a.bn_stride = L.ptr(z), L.ptr(tab), mode, C_, H, W, C_, C_
    a.drop_p, a.drop_seed = p, seed
    return a


def _dz(z, tab, dA, C_, H, W, p=0.0, seed=0):
    d = L.Act()
    d.z, d.bn, d.mode, d.C, d.Hs, d.Ws, d.pstride, d.aux_pstride, d.bn_stride = L.ptr(z), L.ptr(tab), L.ACT_DZ, C_, H, W, C_, C_, C_
    if dA is not None:
        d.aux = L.ptr(dA)
    d.drop_p, d.drop_seed = p, seed
    return d


def _fused(layer, xa0, xa1, g, N, H, W, bwd_of=None, split=0):
    """-> dX [N,H,W,cin] (or the two halves), dW [cout,cin,3,3], backward-sum rows or None"""
    lib = L.load()
    fa = L.FusedBwdArgs()
    fa.xa0, fa.xa1 = xa0, (xa1 if xa1 is not None else L.Act())
    fa.Cin, fa.CinPad, fa.Cout, fa.CoutPad = layer.cin, layer.cin_pad, layer.cout, layer.cout_pad
    d = fa.d
    d.a0, d.a1, d.math, d.wpk = g, L.Act(), L.MATH_BF16X3, L.ptr(layer.wpk16_d)
    d.Cout, d.CoutPad, d.N, d.H, d.W, d.taps = layer.cin, layer.cin_pad, N, H, W, 9
    if split:
        out = torch.full((N, H, W, split), float("nan"), device=DEV)
        out2 = torch.full((N, H, W, layer.cin - split), float("nan"), device=DEV)
        d.out, d.out2, d.out_split, d.out_pstride, d.out2_pstride = L.ptr(out), L.ptr(out2), split, split, layer.cin - split
    else:
        out = torch.full((N, H, W, layer.cin), float("nan"), device=DEV)
        out2 = None
        d.out, d.out_pstride = L.ptr(out), layer.cin
    grid = lib.hpfg_fused_bwd_grid(C.byref(fa))
    assert grid > 0, lib.hpfg_last_error()
    part = None
    if bwd_of is not None:
        part = torch.full((grid, 2, layer.cin_pad), float("nan"), device=DEV)
        d.bwd_stats, d.bwd_of, d.stat_partials = 1, bwd_of, L.ptr(part)
    slab = torch.full((grid, 9, layer.cin_pad, layer.cout_pad), float("nan"), device=DEV)
    fa.slab = L.ptr(slab)
    L.check(lib.hpfg_fused_bwd(C.byref(fa), stream(DEV)), "fused_bwd")
    torch.cuda.synchronize()
    dw = slab.double().sum(0)[:, :layer.cin, :layer.cout].permute(2, 1, 0).reshape(layer.cout, layer.cin, 3, 3).float()
    return (out if out2 is None else (out, out2)), dw, part


CASES = [  # N, H, W, cin, cout, input kind, dZ kind, p_in, p_out
    (2, 32, 48, 16, 16, "bnact", "dz", 0.05, 0.0),     # in_conv.c2 / up4.c2
    (3, 32, 32, 16, 16, "bnact", "dz", 0.0, 0.3),
    (2, 32, 32, 16, 4, "bnact", "plain", 0.0, 0.0),    # out_conv
    (2, 32, 32, 16, 2, "bnact", "plain", 0.0, 0.0),
    (2, 32, 16, 16, 32, "pool", "dz", 0.0, 0.1),       # down1.c1
    (2, 16, 48, 32, 32, "bnact", "dz", 0.1, 0.0),      # down1.c2 / up3.c2
    (2, 32, 32, 32, 16, "cat", "dz", 0.0, 0.0),        # up4.c1
    (2, 32, 32, 32, 16, "cat", "dz", 0.0, 0.0),        # (up3.c1, 64 -> 32, is not instantiated: its weight fragments do not fit LDS)
    (7, 64, 64, 16, 16, "bnact", "dz", 0.0, 0.0),      # more tiles than one round of workgroups x several tiles each
]


@pytest.mark.parametrize("N,H,W,cin,cout,ak,gk,p_in,p_out", CASES)
def test_fused_bwd_equals_separate_kernels_and_autograd(N, H, W, cin, cout, ak, gk, p_in, p_out):
    g = torch.Generator().manual_seed(H * 7 + cin * 3 + cout)
    layer = AdHocConv(cin, cout, 9, DEV, seed=cin + cout, hw=(H, W))
    # ---- layer input
    xa1 = None
    if ak == "bnact":
        zi = torch.randn(N, H, W, cin, generator=g).to(DEV)
        tabi = _bn_table(cin, 3).to(DEV)
        xa0 = _bnact(zi, tabi, cin, H, W, p=p_in, seed=77)
    elif ak == "pool":
        zi = torch.randn(N, 2 * H, 2 * W, cin, generator=g).to(DEV)
        tabi = _bn_table(cin, 3).to(DEV)
        xa0 = _bnact(zi, tabi, cin, 2 * H, 2 * W, mode=L.ACT_BNACT_POOL)
    else:
        c2 = cin // 2
        zi = torch.randn(N, H, W, c2, generator=g).to(DEV)
        tabi = _bn_table(c2, 3).to(DEV)
        xa0 = _bnact(zi, tabi, c2, H, W)
        ud = torch.randn(N, H // 2, W // 2, c2, generator=g).to(DEV)
        xa1 = L.Act()
        xa1.z, xa1.mode, xa1.C, xa1.Hs, xa1.Ws, xa1.pstride = L.ptr(ud), L.ACT_UP2X, c2, H // 2, W // 2, c2
    # ---- dZ source
    if gk == "dz":
        zo = torch.randn(N, H, W, cout, generator=g).to(DEV)
        dA = torch.randn(N, H, W, cout, generator=g).to(DEV)
        tabo = _bn_table(cout, 11).to(DEV)
        gsrc = _dz(zo, tabo, dA, cout, H, W, p=p_out, seed=99)
    else:
        dl = torch.randn(N, H, W, cout, generator=g).to(DEV)
        gsrc = plain_act(dl, cout, H, W)
    # ---- the layer below (whose backward sums the dgrad epilogue produces): only for un-split, un-padded outputs
    bwd_of = None
    if ak == "bnact":
        bwd_of = _dz(zi, tabi, None, cin, H, W, p=p_in, seed=77)
    split = cin // 2 if (ak == "cat" and cin == 32) else 0
    dx, dw, part = _fused(layer, xa0, xa1, gsrc, N, H, W, bwd_of=bwd_of, split=split)
    # ---- the separate launches on the same sources
    dx_ref, _ = layer.conv(gsrc, None, N, H, W, dgrad=True, math=L.MATH_BF16X3)
    dw_ref = layer.wgrad(xa0, xa1, gsrc, N, H, W, math=L.MATH_BF16X3)
    if split:
        dx = torch.cat([dx[0], dx[1]], -1)
    assert torch.equal(dx, dx_ref), f"dgrad differs: {maxerr(dx.cpu(), dx_ref.cpu())}"      # same fragments, same order of MFMAs
    assert maxerr(dw.cpu(), dw_ref.cpu()) < 1e-5 * max(1.0, float(dw_ref.abs().max()))      # other summation order over pixels
    # ---- plain PyTorch fp32 on the CPU
    a_in = _materialize(xa0, xa1, N, H, W, cin)
    dz = _materialize(gsrc, None, N, H, W, cout)
    xr = nchw(a_in).requires_grad_(True)
    wr = layer.w.cpu().clone().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(nchw(dz))
    assert maxerr(nchw(dx.cpu()), xr.grad) < 3e-4 * max(1.0, float(xr.grad.abs().max()))
    assert maxerr(dw.cpu(), wr.grad) < 5e-4 * max(1.0, float(wr.grad.abs().max()))
    if part is not None:
        # sum(g), sum(g * xhat) of the layer below with g = dX * dropout * LeakyReLU'
        lib = L.load()
        bo = _dz(zi, tabi, dx_ref, cin, H, W, p=p_in, seed=77)
        nblk = lib.hpfg_bn_bwd_blocks(N, H, W, cin)
        ref = torch.empty(nblk, 2, cin, device=DEV)
        L.check(lib.hpfg_bn_bwd_reduce(C.byref(bo), N, H, W, L.ptr(ref), stream(DEV)), "bn_bwd_reduce")
        ps, rs = part.double().sum(0).cpu(), ref.double().sum(0).cpu()
        assert maxerr(ps[:, :cin], rs) < 1e-4 * max(1.0, float(rs.abs().max()))


@pytest.mark.parametrize("streaming", [1, 0])
@pytest.mark.parametrize("N,H,W,cin", [(2, 32, 80, 1), (3, 32, 32, 3), (1, 16, 64, 1), (2, 48, 48, 1)])
def test_fused_bwd_weight_gradient_only_first_layer(N, H, W, cin, streaming):
    """The first layer (network input read through its strides, <= 4 channels, nothing to back-propagate): hpfg_fused_bwd without `out`
    = the weight-gradient slabs only, against hpfg_wgrad and PyTorch autograd -- in both forms: the streaming kernel (first_wgrad.hip, exact
    fp32 FMAs; round 5) and the tile kernel it replaces (HPFG_OPT_FIRST_WGRAD = 0, split-bf16 matrix-core products)."""
    prev = L.load().hpfg_set_option(L.OPT_FIRST_WGRAD, streaming)
    try:
        _first_layer_case(N, H, W, cin, streaming)
    finally:
        L.load().hpfg_set_option(L.OPT_FIRST_WGRAD, prev)


def _first_layer_case(N, H, W, cin, streaming):
    g = torch.Generator().manual_seed(cin + H)
    layer = AdHocConv(cin, 16, 9, DEV, seed=9, hw=(H, W))
    x = torch.randn(N, cin, H, W, generator=g).to(DEV)
    xa = L.Act()
    xa.z, xa.mode, xa.C, xa.Hs, xa.Ws = L.ptr(x), L.ACT_STRIDED, cin, H, W
    xa.sn, xa.sc, xa.sy, xa.sx = cin * H * W, H * W, W, 1
    zo = torch.randn(N, H, W, 16, generator=g).to(DEV)
    dA = torch.randn(N, H, W, 16, generator=g).to(DEV)
    tabo = _bn_table(16, 11).to(DEV)
    gsrc = _dz(zo, tabo, dA, 16, H, W, p=0.05, seed=3)
    lib = L.load()
    fa = L.FusedBwdArgs()
    fa.xa0, fa.xa1 = xa, L.Act()
    fa.Cin, fa.CinPad, fa.Cout, fa.CoutPad = cin, 16, 16, 16
    fa.d.a0, fa.d.math, fa.d.Cout, fa.d.CoutPad, fa.d.N, fa.d.H, fa.d.W, fa.d.taps = gsrc, L.MATH_BF16X3, cin, 16, N, H, W, 9
    grid = lib.hpfg_fused_bwd_grid(C.byref(fa))
    assert grid > 0, lib.hpfg_last_error()
    slab = torch.full((grid, 9, 16, 16), float("nan"), device=DEV)
    fa.slab = L.ptr(slab)
    L.check(lib.hpfg_fused_bwd(C.byref(fa), stream(DEV)), "fused_bwd")
    torch.cuda.synchronize()
    dw = slab.double().sum(0)[:, :cin, :16].permute(2, 1, 0).reshape(16, cin, 3, 3).float()
    dw_ref = layer.wgrad(xa, None, gsrc, N, H, W, math=L.MATH_BF16X3)
    assert maxerr(dw.cpu(), dw_ref.cpu()) < (5e-5 if streaming else 1e-5) * max(1.0, float(dw_ref.abs().max()))      # (exact products vs split-bf16 ones)
    dz = _materialize(gsrc, None, N, H, W, 16)
    xr = x.cpu().clone()
    wr = layer.w.cpu().clone().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(nchw(dz))
    assert maxerr(dw.cpu(), wr.grad) < (2e-5 if streaming else 5e-4) * max(1.0, float(wr.grad.abs().max()))


def test_whole_network_fused_and_thin_kernels_equal_the_separate_kernels(monkeypatch):
    """U-Net forward + backward at 64x64 with the whole-tile kernels (conv_thin_kernel, hpfg_fused_bwd) against the same network on the
    chunked forward conv and the separate dgrad + wgrad launches: same logits and parameter gradients up to summation order."""
    from hpfg_amd.model import UNet, reset_dropout_streams

    def run(fused, thin):
        monkeypatch.setenv("HPFG_FUSED_BWD", fused)
        L.load().hpfg_set_option(L.OPT_CONV_THIN, int(thin))      # (the last run below leaves the default, 1, in place)
        reset_dropout_streams()
        torch.manual_seed(3)
        m = UNet(1, 4).to(DEV)
        m.train()
        g = torch.Generator().manual_seed(9)
        x = torch.randn(4, 1, 64, 64, generator=g).to(DEV)
        dy = torch.randn(4, 4, 64, 64, generator=g).to(DEV)
        out = m(x)
        out.backward(dy)
        torch.cuda.synchronize()
        eng = next(iter(m._engines.values()))[0]
        return out.detach().clone(), m.flat_grads.clone(), len(eng.fused_grid)

    o0, g0, n0 = run("0", "0")
    # backward only: the forward pass is bit-identical (same forward kernels), so the gradients differ by summation order alone
    o1, g1, n1 = run("1", "0")
    assert n1 >= 7 and n0 == 0      # in_conv (both convs), down1 (both), up3.c2, up4 (both), out_conv ride the fused kernel at this size
    assert torch.equal(o1, o0)
    assert float((g1 - g0).norm() / g0.norm()) < 2e-5
    # forward too: the raw conv outputs are bit-identical, the BatchNorm sums are partitioned differently (1e-7 on the statistics), and a
    # LeakyReLU sign / max-pool arg-max sitting on a tie may flip: logits tight, gradients to the flip tolerance of tests/test_gpu_unet.py
    o2, g2, _ = run("1", "1")
    assert maxerr(o2.cpu(), o0.cpu()) < 2e-5 * max(1.0, float(o0.abs().max()))
    assert float((g2 - g0).norm() / g0.norm()) < 5e-2
