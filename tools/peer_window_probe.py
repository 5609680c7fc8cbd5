"""What reading a gradient window costs when it lives in uncached (fine-grained) device memory (csrc/peer.hip: hpfg_peer_alloc =
hipExtMallocWithFlags(hipDeviceMallocUncached)) instead of ordinary hipMalloc memory: the owner's pb_reduce_kernel / pb_gather_kernel
stream 7-15 MB out of their own window every step.  One process, one GPU: the streaming kernel hpfg_ema_update
(dst = a * dst + (1 - a) * src, 16 B per lane) with `src` in either kind of memory, sizes of the U-Net / U-Net+ gradient buffers.
usage: python tools/peer_window_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hpfg_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
alpha = torch.full((1,), 0.5, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def time_us(src_ptr, dst, n, reps=200):
    for _ in range(20):
        L.check(lib.hpfg_ema_update(dst.data_ptr(), src_ptr, n, alpha.data_ptr(), st), "ema")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.hpfg_ema_update(dst.data_ptr(), src_ptr, n, alpha.data_ptr(), st), "ema")
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (1_814_000, 3_662_000, 14_650_000):
    dst = torch.zeros(n, device=dev)
    src = torch.randn(n, device=dev)
    p = C.c_void_p()
    L.check(lib.hpfg_peer_alloc(4 * n, C.byref(p)), "peer_alloc")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(p, C.c_void_p(src.data_ptr()), C.c_size_t(4 * n), 3)
    t_c, t_u = time_us(src.data_ptr(), dst, n), time_us(p.value, dst, n)
    gb = 3 * 4 * n / 1e9          # read dst + src, write dst
    print(f"{n} floats ({4 * n / 1e6:.1f} MB window): src in hipMalloc memory {t_c:.1f} us ({gb / t_c * 1e6:.0f} GB/s), src in the uncached window {t_u:.1f} us "
          f"({gb / t_u * 1e6:.0f} GB/s), +{t_u - t_c:.1f} us per pass over the window")
    lib.hpfg_peer_free(p)
