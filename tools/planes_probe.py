"""Diagnostics: launch time of a conv layer's forward kernel with its on-load source vs the same input materialised as split-bf16 planes
(HPFG_ACT_PLANES) -- separates the loader's producer-chain cost from everything else in the kernel.  GPU only."""
import ctypes as C
import os
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


def timeit(fn, reps=20):
    st = torch.cuda.current_stream(dev)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


names = os.environ.get("LAYERS", "encoder.in_conv.conv_conv.4,decoder.up4.conv.conv_conv.4,decoder.up4.conv.conv_conv.0,encoder.down1.maxpool_conv.1.conv_conv.0,"
                       "encoder.down1.maxpool_conv.1.conv_conv.4,decoder.up3.conv.conv_conv.0").split(",")
for name in names:
    s = eng.specs[name]
    a0, a1 = eng.input_acts(name)
    ca = L.ConvArgs()
    ca.math = L.MATH_BF16X3
    ca.wpk = L.ptr(eng.wpk16_f[name])
    ca.bias, ca.out, ca.stat_partials = L.ptr(eng.bias_pad[name]), L.ptr(eng.z[name]), L.ptr(eng.partials)
    ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = s.cout, s.cout, s.cout_pad, eng.N, s.h, s.w, s.taps
    st = torch.cuda.current_stream(dev).cuda_stream
    ca.a0, ca.a1 = a0, a1
    t_on = timeit(lambda: L.check(lib.hpfg_conv_fwd(C.byref(ca), st), "conv"))
    pl = torch.empty(eng.N, s.h, s.w, s.cin, device=dev)
    t_pl = timeit(lambda: L.check(lib.hpfg_act_to_planes(C.byref(a0), C.byref(a1), eng.N, s.h, s.w, L.ptr(pl), st), "planes"))
    ca.a0, ca.a1 = eng._act_planes(pl, s.cin, s.h, s.w), L.Act()
    t_cp = timeit(lambda: L.check(lib.hpfg_conv_fwd(C.byref(ca), st), "conv"))
    mb = eng.N * s.h * s.w * (s.cin + s.cout) * 4 / 1e6
    print(f"{name} ({s.cin}->{s.cout} @{s.h}): on-load {t_on:.1f} us | to_planes {t_pl:.1f} us + conv(planes) {t_cp:.1f} us   [{mb:.0f} MB in+out -> {mb / t_on:.0f} / {mb / t_cp:.0f} GB/ms]")
