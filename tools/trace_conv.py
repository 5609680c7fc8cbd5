"""In-kernel timeline of the bf16x3 conv kernel (diagnostics; needs `make -C hpfg_amd/csrc TRACE=1`).

Wave 0 of every workgroup stamps s_memtime at phase boundaries (ids in conv_bf16_kernel.h):
  1 start, 2 setup done, 3 first tile staged + B ring primed, 4 first barrier passed,
  per chunk: 5 prefetch issued, 6 k-loop done, 7 prefetch finished+parked, 8 barrier passed; 9 tile stored; 10 stats flushed.
Prints, per layer, the mean cycles between consecutive stamp kinds and one workgroup's raw timeline.
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
NAMES = {1: "start", 2: "setup", 3: "stage0", 4: "bar0", 5: "issue", 6: "kloop", 7: "park", 8: "bar", 9: "store", 10: "flush"}


def trace(name):
    s = eng.specs[name]
    a0, a1 = eng.input_acts(name)
    ca = L.ConvArgs()
    ca.a0, ca.a1 = a0, a1
    buf = torch.zeros(8192 * 256, dtype=torch.int64, device=dev)
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
    used = t[:, 0] != 0
    t = t[used]
    ids = (t >> np.uint64(56)).astype(np.int64)
    ts = (t & np.uint64((1 << 56) - 1)).astype(np.int64)
    print(f"== {name} ({s.cin}->{s.cout} @{s.h}): {t.shape[0]} workgroups traced")
    # phase durations: time from previous stamp to this stamp, grouped by this stamp's id
    acc = {}
    life = []
    real = []
    for w in range(t.shape[0]):
        n = int((ids[w] != 0).sum())
        # stamps 11 / 12 carry s_memrealtime (100 MHz, chip-wide); the others s_memtime (shader clock)
        real.append((ts[w, 0], ts[w, n - 1]))
        life.append(ts[w, n - 2] - ts[w, 1])
        for i in range(2, n - 1):
            acc.setdefault(int(ids[w, i]), []).append(int(ts[w, i] - ts[w, i - 1]))
    real = np.array(real)
    span = (real[:, 1].max() - real[:, 0].min()) / 100.0
    lr = (real[:, 1] - real[:, 0]) / 100.0
    print(f"   workgroup lifetime: mean {np.mean(life):.0f} max {np.max(life):.0f} ticks = mean {lr.mean():.2f} us max {lr.max():.2f} us "
          f"(=> {np.mean(life) / lr.mean() / 1e3:.2f} GHz); first start -> last end {span:.2f} us; start spread {(real[:, 0].max() - real[:, 0].min()) / 100.0:.2f} us")
    for k in sorted(acc):
        v = np.array(acc[k])
        print(f"   -> {NAMES.get(k, k):7s}: n/wg {len(v) / t.shape[0]:6.1f}  mean {v.mean():8.0f}  p50 {np.median(v):8.0f}  max {v.max():8.0f}   sum/wg {v.sum() / t.shape[0]:9.0f}")
    w = 0
    n = int((ids[w] != 0).sum())
    print("   wg0:", " ".join(f"{NAMES.get(int(ids[w, i]), '?')}+{int(ts[w, i] - ts[w, i - 1])}" for i in range(2, min(n - 1, 40))))


names = os.environ.get("LAYERS", "encoder.in_conv.conv_conv.4,decoder.up4.conv.conv_conv.0,encoder.down1.maxpool_conv.1.conv_conv.4,"
                       "encoder.down2.maxpool_conv.1.conv_conv.4,encoder.down3.maxpool_conv.1.conv_conv.4,encoder.down4.maxpool_conv.1.conv_conv.4").split(",")
for nm in names:
    trace(nm)
