"""Diagnostics: conv_thin_kernel against conv_bf16x3_kernel on the thin forward layer shapes of the 224x224 U-Net (16 images), HIP-event timed.
usage: python tools/thin_probe.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L  # noqa: E402
from tests.helpers import AdHocConv, stream  # noqa: E402
from tests.test_gpu_kernels import _bn_table  # noqa: E402
from tests.test_gpu_fused_bwd import _bnact  # noqa: E402

DEV = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 16
SHAPES = [("in.c2/up4.c2", 224, 16, 16, "bnact", 0.05), ("out_conv", 224, 16, 4, "bnact", 0.0), ("up4.c1", 224, 32, 16, "cat", 0.0),
          ("d1.c2/up3.c2", 112, 32, 32, "bnact", 0.1), ("down1.c1", 112, 16, 32, "pool", 0.0)]


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
for name, H, cin, cout, ak, p in SHAPES:
    W = H
    layer = AdHocConv(cin, cout, 9, DEV, seed=1, hw=(H, W))
    a1 = L.Act()
    if ak == "bnact":
        z = torch.randn(N, H, W, cin, generator=g).to(DEV)
        tab = _bn_table(cin, 3).to(DEV)
        a0 = _bnact(z, tab, cin, H, W, p=p, seed=7)
        by = 4 * (cin + cout)
    elif ak == "pool":
        z = torch.randn(N, 2 * H, 2 * W, cin, generator=g).to(DEV)
        tab = _bn_table(cin, 3).to(DEV)
        a0 = _bnact(z, tab, cin, 2 * H, 2 * W, mode=L.ACT_BNACT_POOL)
        by = 4 * (4 * cin + cout)
    else:
        c2 = cin // 2
        z = torch.randn(N, H, W, c2, generator=g).to(DEV)
        tab = _bn_table(c2, 3).to(DEV)
        a0 = _bnact(z, tab, c2, H, W)
        ud = torch.randn(N, H // 2, W // 2, c2, generator=g).to(DEV)
        a1.z, a1.mode, a1.C, a1.Hs, a1.Ws, a1.pstride = L.ptr(ud), L.ACT_UP2X, c2, H // 2, W // 2, c2
        by = 4 * (c2 + c2 / 4 + cout)
    out = torch.empty(N, H, W, cout, device=DEV)
    part = torch.zeros(lib.hpfg_conv_stat_blocks(N, H, W) * 2 * layer.cout_pad, device=DEV)
    t = {}
    for tag, math in (("thin", L.MATH_BF16X3), ("chunked", L.MATH_BF16X3 | 0x4000)):
        ca = L.ConvArgs()
        ca.a0, ca.a1, ca.math, ca.wpk, ca.bias, ca.out = a0, a1, math, L.ptr(layer.wpk16_f), L.ptr(layer.bias_pad), L.ptr(out)
        ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = cout, cout, layer.cout_pad, N, H, W, 9
        if cout % 16 == 0:
            ca.stat_partials = L.ptr(part)
        st = stream(DEV)
        t[tag] = timed(lambda: L.check(lib.hpfg_conv_fwd(C.byref(ca), st), "conv"))
        rows = lib.hpfg_conv_stat_rows(C.byref(ca))
    mb = by * N * H * W / 1e6
    print(f"{name:14s} {cin:3d}->{cout:3d} @{H}: thin {t['thin']:6.1f} us ({mb / t['thin'] * 1e3 / 1e3:5.2f} TB/s algorithmic, {rows} workgroups)   chunked {t['chunked']:6.1f} us")
