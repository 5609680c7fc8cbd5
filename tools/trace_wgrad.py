"""In-kernel timeline of the bf16x3 weight-gradient kernel (diagnostics; needs `make -C hpfg_amd/csrc TRACE=1`).

Stamps (wgrad_bf16_kernel.h): 1 start, 2 tables + first batch requested; per work item: 3 barrier passed, 4 A tile parked,
5 dZ tile parked, 6 next item's batch requested, 7 barrier passed, 8 MFMA phase done; 9 end.
One real forward/backward records every layer's wgrad arguments, then each launch is repeated with tracing on.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hpfg_amd import _lib as L

L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libhpfg_hip_trace.so")
from hpfg_amd.model import UNet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
m = UNet(1, 4).to(dev)
m.train()
m.math = "bf16x3"
x = torch.randn(16, 1, 224, 224, device=dev)
lib = L.load()
calls = []
real_wgrad = lib.hpfg_wgrad


class _Spy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        if k == "hpfg_wgrad":
            def f(ref, stream):
                a = L.WgradArgs()
                C.memmove(C.byref(a), ref, C.sizeof(a))
                calls.append(a)
                return real_wgrad(ref, stream)
            return f
        return getattr(self._lib, k)


out = m(x)
eng = next(iter(m._engines.values()))[0]
eng.lib = _Spy(eng.lib)
out.float().square().mean().backward()
torch.cuda.synchronize()
NAMES = {1: "start", 2: "first", 3: "bar1", 4: "parkA", 5: "parkG", 6: "next", 7: "bar2", 8: "mfma", 9: "end"}
want = os.environ.get("SHAPES", "")
buf = torch.zeros(16384 * 256, dtype=torch.int64, device=dev)
for a in calls:
    if a.taps != 9:
        continue
    tag = f"{a.Cin}->{a.Cout}@{a.H}"
    if want and tag not in want.split(","):
        continue
    a.math = L.MATH_BF16X3 | 0x2000
    a.slab = L.ptr(buf)
    a.defer_reduce = 1
    st = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        buf.zero_()
        e0.record(st)
        L.check(real_wgrad(C.byref(a), st.cuda_stream), "wgrad")
        e1.record(st)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 256)
    t = t[t[:, 0] != 0]
    ids = (t >> np.uint64(56)).astype(np.int64)
    ts = (t & np.uint64((1 << 56) - 1)).astype(np.int64)
    acc, life, real = {}, [], []
    for w in range(t.shape[0]):
        n = int((ids[w] != 0).sum())
        real.append((ts[w, 0], ts[w, n - 1]))
        life.append(ts[w, n - 2] - ts[w, 1])
        for i in range(2, n - 1):
            acc.setdefault(int(ids[w, i]), []).append(int(ts[w, i] - ts[w, i - 1]))
    real = np.array(real)
    lr = (real[:, 1] - real[:, 0]) / 100.0
    print(f"== wgrad {tag} S={a.S} modes a0={a.a0.mode} a1={a.a1.mode}: {t.shape[0]} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us; lifetime mean "
          f"{np.mean(life):.0f} ticks = {lr.mean():.2f} us (max {lr.max():.2f}); first start -> last end {(real[:, 1].max() - real[:, 0].min()) / 100.0:.2f} us; "
          f"start spread {(real[:, 0].max() - real[:, 0].min()) / 100.0:.2f} us")
    for k in sorted(acc):
        v = np.array(acc[k])
        print(f"   -> {NAMES.get(k, k):6s}: n/wg {len(v) / t.shape[0]:6.1f}  mean {v.mean():8.0f}  p50 {np.median(v):8.0f}  max {v.max():8.0f}   sum/wg {v.sum() / t.shape[0]:9.0f}")
