"""GPU unit parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch fp32 on the CPU."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hpfg_amd import _lib as L
from oracle import rng_ref
from tests.helpers import AdHocConv, maxerr, nchw, plain_act, stream

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_dropout_mask_matches_cpu_law():
    n, p, seed = 100003, 0.3, 0xDEADBEEF
    out = torch.empty(n, dtype=torch.uint8, device=DEV)
    L.check(L.load().hpfg_dropout_mask(L.ptr(out), n, p, seed, None, stream(DEV)), "mask")
    ref = rng_ref.keep_mask_nhwc(n, p, seed)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert abs(ref.mean() - 0.7) < 0.01


CONV_CASES = [  # N, H, W, cin, cout, taps
    (2, 32, 32, 16, 16, 9), (1, 32, 48, 32, 32, 9), (2, 16, 16, 16, 64, 9), (1, 16, 16, 64, 128, 9),
    (2, 24, 24, 32, 64, 9), (2, 8, 8, 64, 128, 9), (3, 12, 12, 16, 32, 9), (2, 4, 4, 128, 256, 9), (2, 2, 2, 256, 256, 9),
    (2, 6, 6, 32, 16, 9), (2, 32, 32, 16, 4, 9), (2, 32, 32, 16, 2, 9), (2, 32, 32, 4, 16, 9), (2, 16, 16, 2, 16, 9),
    (2, 14, 14, 256, 128, 1), (2, 16, 16, 32, 16, 1), (1, 8, 8, 64, 32, 1), (2, 28, 28, 128, 64, 1),
    # BASELINE-like sizes of the deep layers (16 images): 448 / 256 / 224 workgroups of the 4x16-tile kernel
    (16, 28, 28, 64, 128, 9), (16, 14, 14, 128, 256, 9), (16, 28, 28, 128, 64, 9),
]


@pytest.mark.parametrize("N,H,W,cin,cout,taps", CONV_CASES)
def test_conv_fwd_dgrad_wgrad_plain(N, H, W, cin, cout, taps):
    g = torch.Generator().manual_seed(H * 1000 + cin + cout)
    x = torch.randn(N, H, W, cin, generator=g)
    gz = torch.randn(N, H, W, cout, generator=g)
    layer = AdHocConv(cin, cout, taps, DEV, seed=cin * 7 + cout, hw=(H, W))
    xd, gzd = x.to(DEV), gz.to(DEV)
    out, part = layer.conv(plain_act(xd, cin, H, W), None, N, H, W, stats=True)
    xr = nchw(x).requires_grad_(True)
    wr = layer.w.cpu().clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, layer.b.cpu(), padding=layer.k // 2)
    e = maxerr(nchw(out.cpu()), ref.detach())
    assert e < 2e-4 * max(1.0, float(ref.detach().abs().max())), f"conv fwd err {e}"
    # BN partial sums
    nblk = L.load().hpfg_conv_stat_blocks(N, H, W)
    ps = part.view(nblk, 2, layer.cout_pad).cpu().double().sum(0)
    rs = ref.detach().double()
    assert maxerr(ps[0, :cout], rs.sum((0, 2, 3))) < 1e-2 * max(1.0, float(rs.sum((0, 2, 3)).abs().max()))
    assert maxerr(ps[1, :cout], (rs * rs).sum((0, 2, 3))) < 1e-3 * float((rs * rs).sum((0, 2, 3)).max())
    ref.backward(nchw(gz))
    # dgrad: conv of gz with the transposed / flipped weights
    dx, _ = layer.conv(plain_act(gzd, cout, H, W), None, N, H, W, dgrad=True)
    e = maxerr(nchw(dx.cpu()), xr.grad)
    assert e < 2e-4 * max(1.0, float(xr.grad.abs().max())), f"dgrad err {e}"
    # split-bf16 matrix-core path: fp32-class accuracy (~1e-5 relative), same packing entry point
    o16, p16 = layer.conv(plain_act(xd, cin, H, W), None, N, H, W, stats=True, math=L.MATH_BF16X3)
    e = maxerr(nchw(o16.cpu()), ref.detach())
    assert e < 3e-4 * max(1.0, float(ref.detach().abs().max())), f"bf16x3 conv fwd err {e}"
    d16, _ = layer.conv(plain_act(gzd, cout, H, W), None, N, H, W, dgrad=True, math=L.MATH_BF16X3)
    e = maxerr(nchw(d16.cpu()), xr.grad)
    assert e < 3e-4 * max(1.0, float(xr.grad.abs().max())), f"bf16x3 dgrad err {e}"
    dw = layer.wgrad(plain_act(xd, cin, H, W), None, plain_act(gzd, cout, H, W), N, H, W)
    e = maxerr(dw.cpu(), wr.grad)
    assert e < 3e-4 * max(1.0, float(wr.grad.abs().max())), f"wgrad err {e}"
    if taps == 9:
        dw16 = layer.wgrad(plain_act(xd, cin, H, W), None, plain_act(gzd, cout, H, W), N, H, W, math=L.MATH_BF16X3)
        e = maxerr(dw16.cpu(), wr.grad)
        assert e < 5e-4 * max(1.0, float(wr.grad.abs().max())), f"bf16x3 wgrad err {e}"


def _bn_table(C_, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(L.BN_ROWS, C_)
    t[L.BN_MEAN] = torch.randn(C_, generator=g) * 0.3
    t[L.BN_RSTD] = torch.rand(C_, generator=g) + 0.5
    gamma = torch.randn(C_, generator=g)
    beta = torch.randn(C_, generator=g) * 0.2
    t[L.BN_SCALE] = gamma * t[L.BN_RSTD]
    t[L.BN_SHIFT] = beta - t[L.BN_MEAN] * t[L.BN_SCALE]
    t[L.BN_K1], t[L.BN_K2], t[L.BN_K3] = torch.randn(3, C_, generator=g) * 0.5
    return t


def _materialize(a0, a1, N, H, W, Ct):
    out = torch.full((N, H, W, Ct), float("nan"), device=DEV)
    L.check(L.load().hpfg_act_materialize(C.byref(a0), C.byref(a1) if a1 is not None else None, N, H, W, L.ptr(out), stream(DEV)), "mat")
    return out.cpu()


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_loader_bnact_pool_up_concat_dz(p):
    N, H, W, C_ = 2, 12, 16, 32
    g = torch.Generator().manual_seed(3)
    z = torch.randn(N, H, W, C_, generator=g)
    tab = _bn_table(C_, 5)
    zd, tabd = z.to(DEV), tab.to(DEV)
    seed = 12345
    a = L.Act()
    a.z, a.bn, a.mode, a.C, a.Hs, a.Ws, a.pstride, a.bn_stride = L.ptr(zd), L.ptr(tabd), L.ACT_BNACT, C_, H, W, C_, C_
    a.drop_p, a.drop_seed = p, seed
    y = F.leaky_relu(z * tab[L.BN_SCALE] + tab[L.BN_SHIFT], 0.01)
    keep = torch.ones_like(y)
    if p > 0:
        keep = torch.from_numpy(rng_ref.keep_mask_nhwc(z.numel(), p, seed).astype(np.float32)).view_as(z)
    assert maxerr(_materialize(a, None, N, H, W, C_), y * keep / (1 - p)) < 1e-5
    # pool (no dropout on pooled sources)
    a.mode, a.drop_p = L.ACT_BNACT_POOL, 0.0
    ref = F.max_pool2d(nchw(y), 2)
    assert maxerr(nchw(_materialize(a, None, N, H // 2, W // 2, C_)), ref) < 1e-5
    # concat [bnact | up2x]
    a.mode = L.ACT_BNACT
    u = torch.randn(N, H // 2, W // 2, 16, generator=g)
    ud = u.to(DEV)
    b = L.Act()
    b.z, b.mode, b.C, b.Hs, b.Ws, b.pstride = L.ptr(ud), L.ACT_UP2X, 16, H // 2, W // 2, 16
    up = F.interpolate(nchw(u), scale_factor=2, mode="bilinear", align_corners=True)
    ref = torch.cat([nchw(y), up], 1)
    assert maxerr(nchw(_materialize(a, b, N, H, W, C_ + 16)), ref) < 1e-5
    # dz = k1*g + k2*z + k3 with g = dA*mask/(1-p)*lrelu'
    dA = torch.randn(N, H, W, C_ + 8, generator=g)     # wider tensor: exercises aux_pstride
    dAd = dA.to(DEV)
    d = L.Act()
    d.z, d.bn, d.aux, d.mode, d.C, d.Hs, d.Ws, d.pstride, d.aux_pstride, d.bn_stride = (L.ptr(zd), L.ptr(tabd), L.ptr(dAd), L.ACT_DZ, C_, H, W, C_,
                                                                                       C_ + 8, C_)
    d.drop_p, d.drop_seed = p, seed
    yv = z * tab[L.BN_SCALE] + tab[L.BN_SHIFT]
    gg = dA[..., :C_] * keep / (1 - p) * torch.where(yv > 0, torch.ones_like(yv), torch.full_like(yv, 0.01))
    ref = tab[L.BN_K1] * gg + tab[L.BN_K2] * z + tab[L.BN_K3]
    assert maxerr(_materialize(d, None, N, H, W, C_), ref) < 1e-5
    # BN backward reduction: sum g, sum g*xhat
    lib = L.load()
    nblk = lib.hpfg_bn_bwd_blocks(N, H, W, C_)
    part = torch.empty(nblk * 2 * C_, device=DEV)
    L.check(lib.hpfg_bn_bwd_reduce(C.byref(d), N, H, W, L.ptr(part), stream(DEV)), "bn_bwd_reduce")
    ps = part.view(nblk, 2, C_).cpu().double().sum(0)
    xh = (z - tab[L.BN_MEAN]) * tab[L.BN_RSTD]
    assert maxerr(ps[0], gg.double().sum((0, 1, 2))) < 1e-3
    assert maxerr(ps[1], (gg * xh).double().sum((0, 1, 2))) < 1e-3


def test_bn_finalize_and_running_stats():
    N, H, W, C_ = 3, 16, 16, 32
    g = torch.Generator().manual_seed(9)
    z = torch.randn(N, C_, H, W, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C_)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C_, generator=g))
        bn.bias.copy_(torch.randn(C_, generator=g))
    bn.train()
    y = bn(z)
    zz = z.permute(0, 2, 3, 1).reshape(-1, C_).double()
    part = torch.stack([zz.sum(0), (zz * zz).sum(0)]).float().view(1, 2, C_).to(DEV)
    tab = torch.zeros(L.BN_ROWS, C_, device=DEV)
    rm, rv = torch.zeros(C_, device=DEV), torch.ones(C_, device=DEV)
    gam, bet = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)     # keep alive: the call only enqueues work
    L.check(L.load().hpfg_bn_fwd_finalize(L.ptr(part), 1, None, float(N * H * W), L.ptr(gam), L.ptr(bet),
                                          L.ptr(rm), L.ptr(rv), 0.1, 1e-5, L.ptr(tab), C_, stream(DEV)), "fin")
    t = tab.cpu()
    yk = z * t[L.BN_SCALE].view(1, -1, 1, 1) + t[L.BN_SHIFT].view(1, -1, 1, 1)
    assert maxerr(yk, y.detach()) < 1e-4
    assert maxerr(rm.cpu(), bn.running_mean) < 1e-5 and maxerr(rv.cpu(), bn.running_var) < 1e-4


def test_pool_scatter_and_upsample_bwd():
    N, H, W, C_ = 2, 8, 12, 16
    g = torch.Generator().manual_seed(11)
    z = torch.randn(N, H, W, C_, generator=g)
    tab = _bn_table(C_, 6)
    y = F.leaky_relu(z * tab[L.BN_SCALE] + tab[L.BN_SHIFT], 0.01)
    yr = nchw(y).requires_grad_(True)
    pooled = F.max_pool2d(yr, 2)
    dP = torch.randn(N, H // 2, W // 2, C_, generator=g)
    pooled.backward(nchw(dP))
    base = torch.randn(N, H, W, C_ + 16, generator=g)          # dA lives inside a wider (concat) buffer
    zd, tabd, dPd, dAd = z.to(DEV), tab.to(DEV), dP.to(DEV), base.to(DEV)
    a = L.Act()
    a.z, a.bn, a.mode, a.C, a.Hs, a.Ws, a.pstride, a.bn_stride = L.ptr(zd), L.ptr(tabd), L.ACT_BNACT, C_, H, W, C_, C_
    L.check(L.load().hpfg_pool_scatter_add(C.byref(a), L.ptr(dPd), C_, L.ptr(dAd), C_ + 16, N, H // 2, W // 2, stream(DEV)), "scatter")
    ref = base.clone()
    ref[..., :C_] += yr.grad.permute(0, 2, 3, 1)
    assert maxerr(dAd.cpu(), ref) < 1e-6
    # upsample backward
    for (hl, wl, cu) in [(4, 6, 8), (1, 1, 8), (2, 2, 8), (7, 7, 8), (28, 20, 16), (14, 9, 32), (9, 7, 64), (5, 6, 128)]:
        u = torch.randn(N, cu, hl, wl, generator=g, requires_grad=True)
        up = F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=True)
        gcat = torch.randn(N, 2 * hl, 2 * wl, 16 + cu, generator=g)  # channels [16,16+cu) hold dUp, pixel stride 16+cu
        up.backward(nchw(gcat[..., 16:]))
        gd = gcat.to(DEV)
        dU = torch.empty(N, hl, wl, cu, device=DEV)
        L.check(L.load().hpfg_upsample2x_bwd(gd.view(-1)[16:].data_ptr(), 16 + cu, L.ptr(dU), N, hl, wl, cu, stream(DEV)), "upbwd")
        assert maxerr(nchw(dU.cpu()), u.grad) < 1e-5, (hl, wl, cu)
        rows = L.load().hpfg_upsample2x_bwd_blocks(N, hl, wl, cu)          # fused per-workgroup channel sums (1x1 conv bias gradient)
        part = torch.empty(rows, cu, device=DEV)
        dU2 = torch.empty_like(dU)
        L.check(L.load().hpfg_upsample2x_bwd_sums(gd.view(-1)[16:].data_ptr(), 16 + cu, L.ptr(dU2), N, hl, wl, cu, L.ptr(part), stream(DEV)), "upbwd")
        assert torch.equal(dU2, dU)
        assert maxerr(part.sum(0).cpu(), u.grad.sum((0, 2, 3))) < 1e-3, (hl, wl, cu)


def test_bn_bwd_reduce_pool_equals_scatter_then_reduce():
    """The fused pass (max-pool backward folded into the BatchNorm-backward reduction) against the two separate kernels."""
    lib = L.load()
    for (N, H, W, C_) in [(2, 8, 12, 16), (3, 4, 4, 128), (1, 6, 10, 32)]:
        g = torch.Generator().manual_seed(C_ + H)
        z = torch.randn(N, H, W, C_, generator=g).to(DEV)
        tab = _bn_table(C_, 9).to(DEV)
        dP = torch.randn(N, H // 2, W // 2, C_, generator=g).to(DEV)
        base = torch.randn(N, H, W, C_ + 16, generator=g).to(DEV)      # dA is the first C_ channels of a wider (concat) buffer
        two, one = base.clone(), base.clone()

        def act(dA):
            a = L.Act()
            a.z, a.bn, a.aux, a.mode, a.C = L.ptr(z), L.ptr(tab), L.ptr(dA), L.ACT_DZ, C_
            a.Hs, a.Ws, a.pstride, a.aux_pstride, a.bn_stride = H, W, C_, C_ + 16, C_
            return a
        src = L.Act()
        src.z, src.bn, src.mode, src.C, src.Hs, src.Ws, src.pstride, src.bn_stride = L.ptr(z), L.ptr(tab), L.ACT_BNACT, C_, H, W, C_, C_
        L.check(lib.hpfg_pool_scatter_add(C.byref(src), L.ptr(dP), C_, L.ptr(two), C_ + 16, N, H // 2, W // 2, stream(DEV)), "scatter")
        nb2 = lib.hpfg_bn_bwd_blocks(N, H, W, C_)
        p2 = torch.zeros(nb2, 2, C_, device=DEV)
        L.check(lib.hpfg_bn_bwd_reduce(C.byref(act(two)), N, H, W, L.ptr(p2), stream(DEV)), "reduce")
        nb1 = lib.hpfg_bn_bwd_pool_blocks(N, H // 2, W // 2, C_)
        p1 = torch.zeros(nb1, 2, C_, device=DEV)
        L.check(lib.hpfg_bn_bwd_reduce_pool(C.byref(act(one)), L.ptr(dP), C_, N, H // 2, W // 2, L.ptr(p1), stream(DEV)), "fused")
        assert torch.equal(one, two)                                    # same arg-max rule, same in-place sum
        assert maxerr(p1.sum(0).cpu(), p2.sum(0).cpu()) < 1e-3 * max(1.0, float(p2.sum(0).abs().max()))


def test_first_conv_and_channel_sum():
    lib = L.load()
    for (N, H, W, cin) in [(2, 32, 32, 1), (2, 16, 48, 3)]:
        g = torch.Generator().manual_seed(cin)
        x = torch.randn(N, cin, H, W, generator=g)
        w = torch.randn(16, cin, 3, 3, generator=g) * 0.3
        b = torch.randn(16, generator=g)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        a = L.Act()
        a.z, a.mode, a.C, a.Hs, a.Ws = L.ptr(xd), L.ACT_STRIDED, cin, H, W
        a.sn, a.sc, a.sy, a.sx = xd.stride()
        out = torch.empty(N, H, W, 16, device=DEV)
        nblk = lib.hpfg_conv_first_rows(N, H, W)
        part = torch.empty(nblk * 2 * 16, device=DEV)
        L.check(lib.hpfg_conv3x3_first_fwd(C.byref(a), L.ptr(wd), L.ptr(bd), L.ptr(out), L.ptr(part), N, H, W, cin, 16, stream(DEV)), "first")
        ref = F.conv2d(x, w, b, padding=1)
        assert maxerr(nchw(out.cpu()), ref) < 1e-4
        ps = part.view(nblk, 2, 16).cpu().double().sum(0)
        assert maxerr(ps[0], ref.double().sum((0, 2, 3))) < 1e-2
        cs = torch.empty(16, device=DEV)
        scratch = torch.empty(512 * 256, device=DEV)
        L.check(lib.hpfg_channel_sum(L.ptr(out), 16, N * H * W, 16, L.ptr(cs), L.ptr(scratch), stream(DEV)), "csum")
        assert maxerr(cs.cpu(), ref.sum((0, 2, 3))) < 1e-2


def test_box_masks_device_equals_host():
    """CutMix masks rasterised on the device are the host (reference-law) masks for the same draws, incl. numpy's slice semantics
    for out-of-range boxes (within_bounds=False)."""
    import numpy as np
    from hpfg_amd.utils.utils import BoxMaskGenerator
    for kw in (dict(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=True),
               dict(prop_range=(0.1, 0.9), n_boxes=3, random_aspect_ratio=True, prop_by_area=False, within_bounds=False, invert=False)):
        gen = BoxMaskGenerator(**kw)
        for shape in ((224, 224), (37, 53)):
            host = gen.generate_params(6, shape, rng=np.random.RandomState(11))
            dev = gen.generate_params_device(6, shape, DEV, rng=np.random.RandomState(11))
            assert dev.shape == host.shape and np.array_equal(dev.cpu().numpy(), host.astype(np.float32))


def test_sgd_ema_step_equals_the_two_launches_bitwise():
    """hpfg_sgd_ema_step == hpfg_sgd_step followed by hpfg_ema_update (full buffer and a leading slice), bit for bit."""
    from hpfg_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    n = 100_003
    st = torch.cuda.current_stream(DEV).cuda_stream
    for n_ema in (n, 4097, 0):
        p0, gr, m0, t0 = (torch.randn(n, generator=g).to(DEV) for _ in range(4))
        lr = torch.tensor([0.0123], device=DEV)
        al = torch.tensor([0.987], device=DEV)
        pa, ma, ta = p0.clone(), m0.clone(), t0.clone()
        L.check(lib.hpfg_sgd_step(L.ptr(pa), L.ptr(gr), L.ptr(ma), n, L.ptr(lr), 0.9, 5e-4, 0.5, st), "sgd")
        if n_ema:
            L.check(lib.hpfg_ema_update(L.ptr(ta), L.ptr(pa), n_ema, L.ptr(al), st), "ema")
        pb, mb, tb = p0.clone(), m0.clone(), t0.clone()
        L.check(lib.hpfg_sgd_ema_step(L.ptr(pb), L.ptr(gr), L.ptr(mb), n, L.ptr(lr), 0.9, 5e-4, 0.5, L.ptr(tb), n_ema, L.ptr(al), st), "sgd_ema")
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(ta, tb), n_ema


def test_forward_advances_bn_counters_and_seed_word_inside_the_pack_launch():
    """num_batches_tracked += 1 per train-mode forward and the engine's dropout seed word (host-counted when eager, +1 on the device in
    graph-seed mode) are advanced by hpfg_pack_weights_bump, the forward's first launch -- same values as nn.BatchNorm2d would hold."""
    from hpfg_amd.model import UNet
    from hpfg_amd.model.unet import reset_dropout_streams
    reset_dropout_streams()
    torch.manual_seed(0)
    m = UNet(1, 4).to(DEV)
    m.train()
    x = torch.randn(2, 1, 32, 32, device=DEV)
    for k in range(1, 4):
        m(x).sum().backward()                          # (backward releases the engine: the next forward takes the same one)
        torch.cuda.synchronize()
        nbt = [int(b) for n, b in m.named_buffers() if n.endswith("num_batches_tracked")]
        assert len(nbt) == 18 and all(v == k for v in nbt), (k, nbt)
    engs = [e for pool in m._engines.values() for e in pool]
    assert len(engs) == 1
    eng = engs[0]
    assert int(eng.seed_dev) == 3                      # eager: the module's per-forward counter
    m._graph_seed_mode = True                          # what GraphedStep sets while capturing: the forward bumps the word on the device
    m(x).sum().backward()
    m(x).sum().backward()
    m._graph_seed_mode = False
    torch.cuda.synchronize()
    assert int(eng.seed_dev) == 5
    m.eval()
    with torch.no_grad():
        m(x)
    torch.cuda.synchronize()
    assert all(int(b) == 5 for n, b in m.named_buffers() if n.endswith("num_batches_tracked"))      # eval forwards do not count


@pytest.mark.parametrize("cin", [1, 3])
@pytest.mark.parametrize("N,H,W", [(2, 32, 32), (3, 224, 224), (1, 16, 48), (2, 40, 24), (4, 96, 96)])
def test_first_conv_mfma_form_equals_the_valu_form(N, H, W, cin, monkeypatch):
    """The 1- / 3-channel first layer on v_mfma_f32_16x16x4_f32 (three / seven k-steps of 4 taps: an fmaf chain in (channel, tap) order) gives the output and the
    BatchNorm partial sums of the VALU loop it replaces, bit for bit (the fixture traces of tests/test_gpu_steps.py are chaotic enough on their
    4-image batches that a last-bit change of the first layer's statistics moves the logits by 1e-3 after three steps); also on sizes that are
    not multiples of 16 (no statistics there)."""
    lib = L.load()
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, cin, H, W, generator=g).to(DEV)
    w = (torch.randn(16, cin, 3, 3, generator=g) * 0.3).to(DEV)
    b = torch.randn(16, generator=g).to(DEV)
    a = L.Act()
    a.z, a.mode, a.C, a.Hs, a.Ws = L.ptr(x), L.ACT_STRIDED, cin, H, W
    a.sn, a.sc, a.sy, a.sx = x.stride()
    nblk = lib.hpfg_conv_first_rows(N, H, W)
    with_stats = H % 16 == 0 and W % 16 == 0
    res = []
    for form in ("0", "1"):
        lib.hpfg_set_option(L.OPT_FIRST_MFMA, int(form))
        out = torch.full((N, H, W, 16), float("nan"), device=DEV)
        part = torch.full((nblk * 2 * 16,), float("nan"), device=DEV)
        L.check(lib.hpfg_conv3x3_first_fwd(C.byref(a), L.ptr(w), L.ptr(b), L.ptr(out), L.ptr(part) if with_stats else None, N, H, W, cin, 16, stream(DEV)), "first")
        torch.cuda.synchronize()
        res.append((out, part.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert maxerr(nchw(res[1][0].cpu()), F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)) < 1e-4
    if with_stats:
        assert torch.equal(res[0][1], res[1][1])          # the partial-sum rows too: same association of the in-workgroup sums
