"""Diagnostics: hpfg_fused_bwd against the separate hpfg_wgrad + hpfg_conv_fwd(dgrad) pair on the thin layer shapes of the 224x224 U-Net
(16 images), HIP-event timed.  usage: python tools/fused_probe.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L  # noqa: E402
from tests.helpers import AdHocConv, plain_act, stream  # noqa: E402
from tests.test_gpu_kernels import _bn_table  # noqa: E402
from tests.test_gpu_fused_bwd import _bnact, _dz  # noqa: E402

DEV = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 16
SHAPES = [  # name, H, cin, cout, input kind, dZ kind, stats
    ("out_conv", 224, 16, 4, "bnact", "plain", True), ("up4.c2/in.c2", 224, 16, 16, "bnact", "dz", True), ("up4.c1", 224, 32, 16, "cat", "dz", False),
    ("up3.c2/d1.c2", 112, 32, 32, "bnact", "dz", True), ("down1.c1", 112, 16, 32, "pool", "dz", False),
]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


lib = L.load()
g = torch.Generator().manual_seed(0)
for name, H, cin, cout, ak, gk, stats in SHAPES:
    W = H
    layer = AdHocConv(cin, cout, 9, DEV, seed=1, hw=(H, W))
    xa1 = None
    if ak == "bnact":
        zi = torch.randn(N, H, W, cin, generator=g).to(DEV)
        tabi = _bn_table(cin, 3).to(DEV)
        xa0 = _bnact(zi, tabi, cin, H, W, p=0.05 if cin == 16 else 0.0, seed=7)
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
    if gk == "dz":
        zo = torch.randn(N, H, W, cout, generator=g).to(DEV)
        dA = torch.randn(N, H, W, cout, generator=g).to(DEV)
        tabo = _bn_table(cout, 11).to(DEV)
        gsrc = _dz(zo, tabo, dA, cout, H, W)
    else:
        dl = torch.randn(N, H, W, cout, generator=g).to(DEV)
        gsrc = plain_act(dl, cout, H, W)
    out = torch.empty(N, H, W, cin, device=DEV)
    fa = L.FusedBwdArgs()
    fa.xa0, fa.xa1 = xa0, (xa1 if xa1 is not None else L.Act())
    fa.Cin, fa.CinPad, fa.Cout, fa.CoutPad = cin, layer.cin_pad, cout, layer.cout_pad
    d = fa.d
    d.a0, d.math, d.wpk, d.out, d.out_pstride = gsrc, L.MATH_BF16X3, L.ptr(layer.wpk16_d), L.ptr(out), cin
    d.Cout, d.CoutPad, d.N, d.H, d.W, d.taps = cin, layer.cin_pad, N, H, W, 9
    grid = lib.hpfg_fused_bwd_grid(C.byref(fa))
    part = torch.empty(max(grid, lib.hpfg_conv_stat_blocks(N, H, W)) * 2 * layer.cin_pad, device=DEV)
    bo = None
    if stats:
        bo = _dz(zi, tabi, None, cin, H, W, p=0.05 if cin == 16 else 0.0, seed=7)
        d.bwd_stats, d.bwd_of, d.stat_partials = 1, bo, L.ptr(part)
    slab = torch.empty(grid * 9 * layer.cin_pad * layer.cout_pad, device=DEV)
    fa.slab = L.ptr(slab)
    st = stream(DEV)
    t_f = timed(lambda: L.check(lib.hpfg_fused_bwd(C.byref(fa), st), "fused"))
    # the separate pair
    ca = L.ConvArgs()
    ca.a0, ca.a1, ca.math, ca.wpk, ca.out, ca.out_pstride = gsrc, L.Act(), L.MATH_BF16X3, L.ptr(layer.wpk16_d), L.ptr(out), cin
    ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = cin, layer.cin_pad, N, H, W, 9
    spills = layer.cin_pad % 32 == 0
    if stats and not spills:
        ca.bwd_stats, ca.bwd_of, ca.stat_partials = 1, bo, L.ptr(part)
    wa = L.WgradArgs()
    wa.a0, wa.a1, wa.g = fa.xa0, fa.xa1, gsrc
    S = lib.hpfg_wgrad_splits(N, H, W, layer.cin_pad, layer.cout_pad, 9)
    slab2 = torch.empty(lib.hpfg_wgrad_slab_floats(N, H, W, layer.cin_pad, layer.cout_pad, 9), device=DEV)
    dw = torch.empty_like(layer.w)
    wa.slab, wa.dw_oihw, wa.math, wa.defer_reduce = L.ptr(slab2), L.ptr(dw), L.MATH_BF16X3, 1
    wa.Cin, wa.CinPad, wa.Cout, wa.CoutPad, wa.N, wa.H, wa.W, wa.taps, wa.S = cin, layer.cin_pad, cout, layer.cout_pad, N, H, W, 9, S
    t_d = timed(lambda: L.check(lib.hpfg_conv_fwd(C.byref(ca), st), "dgrad"))
    t_w = timed(lambda: L.check(lib.hpfg_wgrad(C.byref(wa), st), "wgrad"))
    # bytes one pass must move: dA + z of the layer (or dlogits), the layer input once (+ z of the layer below when it is another tensor), dX
    px = N * H * W
    by = px * 4 * ((2 * cout if gk == "dz" else cout) + cin + cin) + (px * 4 * cin if False else 0)
    print(f"{name:14s} {cin:3d}->{cout:3d} @{H}: fused {t_f:6.1f} us (grid {grid}, {by / t_f / 1e3:6.0f} GB/s algorithmic)   dgrad {t_d:6.1f} + wgrad {t_w:6.1f} = {t_d + t_w:6.1f} us")
