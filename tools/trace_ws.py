"""Timeline of the warp-specialised conv kernel (diagnostics; needs `make -C hpfg_amd/csrc TRACE=1`).

Per workgroup two records: MMA wave 0 (entries 0..127) and loader wave 4 (128..255).  Stamp ids (conv_ws_kernel.h):
MMA: 1 start, 2 tables in LDS, 4 ring primed, 5 image 0 ready, per step: 6 k-loop done, 8 barrier passed, 9 tile stored; 10 end.
loader: 3 first positions requested, 4 position 0 parked, 5 barrier, per step: 6 parked, 7 next requested, 8 barrier passed; 10 end.
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
with torch.no_grad():
    m(x)
eng = next(iter(m._engines.values()))[0]
lib = L.load()
MN = {1: "start", 2: "tables", 4: "primed", 5: "img0", 6: "kloop", 8: "bar", 9: "store", 10: "end"}
LN = {1: "start", 2: "tables", 3: "req0", 4: "park0", 5: "bar0", 6: "park", 7: "req", 8: "bar", 10: "end"}


def summarize(t, names, label):
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
    print(f"   [{label}] {t.shape[0]} records; lifetime mean {lr.mean():.2f} us max {lr.max():.2f} us; first start -> last end "
          f"{(real[:, 1].max() - real[:, 0].min()) / 100.0:.2f} us")
    for k in sorted(acc):
        v = np.array(acc[k])
        print(f"      -> {names.get(k, k):7s}: n/wg {len(v) / t.shape[0]:6.1f}  mean {v.mean():8.0f}  p50 {np.median(v):8.0f}  max {v.max():8.0f}   sum/wg {v.sum() / t.shape[0]:9.0f}")


def trace(name):
    s = eng.specs[name]
    a0, a1 = eng.input_acts(name)
    ca = L.ConvArgs()
    ca.a0, ca.a1 = a0, a1
    buf = torch.zeros(4096 * 256, dtype=torch.int64, device=dev)
    ca.math = L.MATH_BF16X3 | 0x1000 | 0x2000
    ca.wpk = L.ptr(eng.wpk16_f[name])
    ca.bias, ca.out, ca.stat_partials = L.ptr(eng.bias_pad[name]), L.ptr(eng.z[name]), L.ptr(buf)
    ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = s.cout, s.cout, s.cout_pad, eng.N, s.h, s.w, s.taps
    st = torch.cuda.current_stream(dev)
    for _ in range(2):
        buf.zero_()
        L.check(lib.hpfg_conv_fwd(C.byref(ca), st.cuda_stream), "conv")
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 256)
    print(f"== {name} ({s.cin}->{s.cout} @{s.h})")
    summarize(t[:, :128], MN, "MMA wave 0")
    summarize(t[:, 128:], LN, "loader wave 4")


names = os.environ.get("LAYERS", "encoder.in_conv.conv_conv.4,decoder.up4.conv.conv_conv.0,encoder.down3.maxpool_conv.1.conv_conv.4").split(",")
for nm in names:
    trace(nm)
