"""In-kernel timeline of hpfg_fused_bwd (diagnostics; needs `make -C hpfg_amd/csrc TRACE=1`).

Thread 0 of every workgroup stamps s_memtime at the phase boundaries (ids in fused_bwd_kernel.h):
  1 start, 2 tables / weight fragments in place, per tile: 3 top (late loads requested), 4 staging converted and written, 5 barrier passed,
  6 dgrad MFMAs done, 7 epilogue done, 8 prefetch issued, 9 wgrad MFMAs done, 10 barrier passed; 11 loop done.
Prints the mean cycles between consecutive stamps per layer shape and one workgroup's raw timeline.
usage: python tools/trace_fused.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libhpfg_hip_trace.so")
from tests.helpers import AdHocConv, plain_act, stream  # noqa: E402
from tests.test_gpu_kernels import _bn_table  # noqa: E402
from tests.test_gpu_fused_bwd import _bnact, _dz  # noqa: E402

DEV = torch.device("cuda:0")
N = 16
SHAPES = [("out_conv", 224, 16, 4, "bnact", "plain", True), ("up4.c2/in.c2", 224, 16, 16, "bnact", "dz", True), ("up4.c1", 224, 32, 16, "cat", "dz", False),
          ("up3.c2/d1.c2", 112, 32, 32, "bnact", "dz", True), ("down1.c1", 112, 16, 32, "pool", "dz", False)]
NAMES = {1: "start", 2: "setup", 3: "top", 4: "finish", 5: "bar", 6: "dgrad", 7: "epi", 8: "issue", 9: "wgrad", 10: "bar2", 11: "end"}
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
    d.a0, d.math, d.wpk, d.out, d.out_pstride = gsrc, L.MATH_BF16X3 | 0x2000, L.ptr(layer.wpk16_d), L.ptr(out), cin
    d.Cout, d.CoutPad, d.N, d.H, d.W, d.taps = cin, layer.cin_pad, N, H, W, 9
    grid = lib.hpfg_fused_bwd_grid(C.byref(fa))
    part = torch.empty(grid * 2 * layer.cin_pad, device=DEV)
    if stats:
        d.bwd_stats, d.bwd_of, d.stat_partials = 1, _dz(zi, tabi, None, cin, H, W, p=0.05 if cin == 16 else 0.0, seed=7), L.ptr(part)
    slab = torch.empty(grid * 9 * layer.cin_pad * layer.cout_pad, device=DEV)
    fa.slab = L.ptr(slab)
    buf = torch.zeros(grid * 256, dtype=torch.int64, device=DEV)
    d.bias = L.ptr(buf)
    for _ in range(2):
        buf.zero_()
        L.check(lib.hpfg_fused_bwd(C.byref(fa), stream(DEV)), "fused")
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 256)
    t = t[t[:, 0] != 0]
    ids = (t >> np.uint64(56)).astype(np.int64)
    ts = (t & np.uint64((1 << 56) - 1)).astype(np.int64)
    acc, real = {}, []
    for w in range(t.shape[0]):
        n = int((ids[w] != 0).sum())
        real.append((ts[w, 0], ts[w, n - 1]))
        for i in range(2, n - 1):
            acc.setdefault(int(ids[w, i]), []).append(int(ts[w, i] - ts[w, i - 1]))
    real = np.array(real)
    lr = (real[:, 1] - real[:, 0]) / 100.0
    print(f"== {name} ({cin}->{cout} @{H}): {t.shape[0]} workgroups; lifetime mean {lr.mean():.2f} us max {lr.max():.2f} us; first start -> last end "
          f"{(real[:, 1].max() - real[:, 0].min()) / 100.0:.2f} us; start spread {(real[:, 0].max() - real[:, 0].min()) / 100.0:.2f} us")
    for k in sorted(acc):
        v = np.array(acc[k])
        print(f"   -> {NAMES.get(k, k):7s}: n/wg {len(v) / t.shape[0]:6.1f}  mean {v.mean():8.0f}  p50 {np.median(v):8.0f}  max {v.max():8.0f}   sum/wg {v.sum() / t.shape[0]:9.0f}")
    n = int((ids[0] != 0).sum())
    print("   wg0:", " ".join(f"{NAMES.get(int(ids[0, i]), '?')}+{int(ts[0, i] - ts[0, i - 1])}" for i in range(2, min(n - 1, 36))))
