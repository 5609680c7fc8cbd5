import ctypes as C, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
N, H, W, Cin = 16, 224, 224, 1
x = torch.randn(N, Cin, H, W, device=dev)
w = torch.randn(16, Cin, 3, 3, device=dev) * 0.3
b = torch.randn(16, device=dev)
a = L.Act()
a.z, a.mode, a.C, a.Hs, a.Ws = L.ptr(x), L.ACT_STRIDED, Cin, H, W
a.sn, a.sc, a.sy, a.sx = Cin * H * W, H * W, W, 1
rows = lib.hpfg_conv_first_rows(N, H, W)
outs, parts = [], []
for r in range(3):
    out = torch.full((N, H, W, 16), float("nan"), device=dev)
    part = torch.full((rows, 2, 16), float("nan"), device=dev)
    L.check(lib.hpfg_conv3x3_first_fwd(C.byref(a), L.ptr(w), L.ptr(b), L.ptr(out), L.ptr(part), N, H, W, Cin, 16, torch.cuda.current_stream().cuda_stream), "first")
    torch.cuda.synchronize()
    outs.append(out); parts.append(part)
ref = torch.nn.functional.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1)
print("out equal:", torch.equal(outs[0], outs[1]), torch.equal(outs[1], outs[2]), "max err vs torch", float((outs[0] - ref).abs().max()))
print("part equal:", torch.equal(parts[0], parts[1]), torch.equal(parts[1], parts[2]), "nan rows", int(torch.isnan(parts[0]).any(dim=(1, 2)).sum()))
print("sum check", float((parts[0][:, 0].double().sum(0).cpu() - ref.double().sum((0, 1, 2)).cpu()).abs().max()))
d = (parts[0] - parts[1]).abs()
print("diff rows", d.amax(dim=(1, 2)).nonzero().flatten()[:10].tolist(), float(d.max()))
