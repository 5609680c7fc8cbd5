"""Diagnostics: time hpfg_gemm_bf16x3 / hpfg_gemm_tn_bf16x3 on the SegFormer layer shapes (GPU only)."""
import sys
import torch
sys.path.insert(0, ".")
from hpfg_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.load()
st = torch.cuda.current_stream(dev).cuda_stream


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [(100352, 256, 1024), (100352, 1024, 256), (100352, 128, 32), (100352, 32, 128), (25088, 256, 64), (25088, 64, 256), (6272, 640, 160), (1568, 1024, 256),
          (100352, 32, 32), (100352, 4, 256)]
for R, N, K in shapes:
    x = torch.randn(R, K, device=dev)
    w = torch.randn(N, K, device=dev)
    dy = torch.randn(R, N, device=dev)
    y = torch.empty(R, N, device=dev)
    dx = torch.empty(R, K, device=dev)
    dw = torch.empty(N, K, device=dev)
    part = torch.empty(lib.hpfg_gemm_tn_splits(R, N, K) * N * K, device=dev)
    t_f = timeit(lambda: L.check(lib.hpfg_gemm_bf16x3(L.ptr(x), K, 1, L.ptr(w), 1, K, L.ptr(y), N, R, N, K, None, 0, 0, st), "f"))
    t_d = timeit(lambda: L.check(lib.hpfg_gemm_bf16x3(L.ptr(dy), N, 1, L.ptr(w), K, 1, L.ptr(dx), K, R, K, N, None, 0, 0, st), "d"))
    t_w = timeit(lambda: L.check(lib.hpfg_gemm_tn_bf16x3(L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(part), R, N, K, 0, st), "w"))
    t_t = timeit(lambda: torch.matmul(x, w.t()))
    gf = 2.0 * R * N * K / 1e9
    print(f"R={R} N={N} K={K} ({gf:.1f} GFLOP): fwd {t_f:.0f} us ({gf / t_f * 1e3:.0f} TF)  dX {t_d:.0f} us ({gf / t_d * 1e3:.0f} TF)  dW {t_w:.0f} us ({gf / t_w * 1e3:.0f} TF) "
          f"| torch fp32 matmul fwd {t_t:.0f} us ({gf / t_t * 1e3:.0f} TF)")
