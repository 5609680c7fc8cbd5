"""Run each listed conv layer once per math mode (for rocprofv3 --pmc dynamic instruction counts)."""
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
for name in sys.argv[1:]:
    s = eng.specs[name]
    a0, a1 = eng.input_acts(name)
    ca = L.ConvArgs()
    ca.a0, ca.a1 = a0, a1
    ca.math = L.MATH_BF16X3
    ca.wpk = L.ptr(eng.wpk16_f[name])
    ca.bias, ca.out, ca.stat_partials = L.ptr(eng.bias_pad[name]), L.ptr(eng.z[name]), L.ptr(eng.partials)
    ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = s.cout, s.cout, s.cout_pad, eng.N, s.h, s.w, s.taps
    for _ in range(3):
        L.check(lib.hpfg_conv_fwd(C.byref(ca), torch.cuda.current_stream(dev).cuda_stream), "conv")
torch.cuda.synchronize()
