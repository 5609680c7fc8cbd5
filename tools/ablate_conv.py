"""Per-layer timing of the conv kernels of the U-Net (GPU only, diagnostics).  QUICK=1: just the launch time of each layer in LAYERS
(default kernels).  Without QUICK: a coarse ablation through the
HpfgConvArgs.math debug bits that still exist -- 0x100 no output stores, 0x400 staged pieces forced to zero (the loads are
still issued since the staging became branch-free), 0x1000 no BatchNorm partial sums; 0x200 / 0x800 are no-ops kept for old logs.
For where the time goes inside a launch use tools/trace_conv.py (in-kernel timeline) instead."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from hpfg_amd import _lib as L
from hpfg_amd.model import UNet

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


def run(name, flags, math=L.MATH_BF16X3, reps=20):
    s = eng.specs[name]
    a0, a1 = eng.input_acts(name)
    ca = L.ConvArgs()
    ca.a0, ca.a1 = a0, a1
    ca.math = math | flags
    ca.wpk = L.ptr(eng.wpk16_f[name]) if math == L.MATH_BF16X3 else L.ptr(eng.wpk_f[name])
    ca.bias, ca.out, ca.stat_partials = L.ptr(eng.bias_pad[name]), L.ptr(eng.z[name]), L.ptr(eng.partials)
    ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = s.cout, s.cout, s.cout_pad, eng.N, s.h, s.w, s.taps
    st = torch.cuda.current_stream(dev)
    for _ in range(3):
        L.check(lib.hpfg_conv_fwd(C.byref(ca), st.cuda_stream), "conv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        L.check(lib.hpfg_conv_fwd(C.byref(ca), st.cuda_stream), "conv")
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


import os
names = os.environ.get("LAYERS", "encoder.in_conv.conv_conv.4,decoder.up4.conv.conv_conv.0,encoder.down1.maxpool_conv.1.conv_conv.4,"
                       "encoder.down3.maxpool_conv.1.conv_conv.4,encoder.down4.maxpool_conv.1.conv_conv.4").split(",")
if os.environ.get("QUICK"):
    for name in names:
        s = eng.specs[name]
        print(f"{name} ({s.cin}->{s.cout} @{s.h}): {run(name, int(os.environ.get('FLAGS', '0'), 0)):.1f} us")
    raise SystemExit
for name in names:
    s = eng.specs[name]
    base = run(name, 0)
    f32 = run(name, 0, L.MATH_F32)
    row = {"full": base, "f32": f32, "no_store": run(name, 0x100), "no_mfma": run(name, 0x200), "no_tile_loads": run(name, 0x400), "no_B": run(name, 0x800),
           "no_stats": run(name, 0x1000), "only_loads": run(name, 0x100 | 0x200 | 0x800 | 0x1000), "only_mfma": run(name, 0x100 | 0x400 | 0x800 | 0x1000),
           "nothing": run(name, 0x100 | 0x200 | 0x400 | 0x800 | 0x1000)}
    gf = eng.N * s.h * s.w * s.cin * s.cout * 18 / 1e9
    print(f"{name} ({s.cin}->{s.cout} @{s.h}) GFLOP={gf:.2f}: " + "  ".join(f"{k}={v:.1f}us" for k, v in row.items()))
