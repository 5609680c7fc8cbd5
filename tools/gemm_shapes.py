"""Which GEMM shapes the SegFormer branch of one CTCT step launches and what each costs (diagnostics, GPU only).
Records every hpfg_gemm_bf16x3 / hpfg_gemm_tn_bf16x3 call of one eager step (shape + operand orientation), then times each distinct call alone
(20 launches, HIP events) and prints them by total time per step."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd import _lib as L  # noqa: E402
from hpfg_amd import ops_tokens as T  # noqa: E402
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import build_model  # noqa: E402
from hpfg_amd.train import CTCTStep  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = loadyaml(os.path.join(ROOT, "config", "ctct_unet_segformer_30k_224x224_ACDC.yaml"))
torch.manual_seed(a.seed)
m1, m2 = build_model(a.model1).to(DEV), build_model(a.model2).to(DEV)
m1.train(), m2.train()
xl, yl = synth_batch(10, 8, 224, 224, 1, 4, 32)
xu, _ = synth_batch(11, 24, 224, 224, 1, 4, 32)
xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
st = CTCTStep(m1, m2, a)
st.step(xl, yl, xu, 1)
calls = collections.Counter()
orig = T._gemm


def rec(a_, sam, sak, b_, sbk, sbn, m, n, k, bias=None, math="bf16x3"):
    calls[("gemm", m, n, k, "A k-fast" if sak == 1 else "A m-fast", "B k-fast" if sbk == 1 else "B n-fast")] += 1
    return orig(a_, sam, sak, b_, sbk, sbn, m, n, k, bias, math)


T._gemm = rec
lib = L.load()
orig_tn = lib.hpfg_gemm_tn_bf16x3


def rec_tn(dy, x, wb, part, R, N, K, has_b, stream):
    calls[("gemm_tn", R, N, K, "", "")] += 1
    return orig_tn(dy, x, wb, part, R, N, K, has_b, stream)


lib.hpfg_gemm_tn_bf16x3 = rec_tn
st.step(xl, yl, xu, 2)
torch.cuda.synchronize()
T._gemm = orig
lib.hpfg_gemm_tn_bf16x3 = orig_tn
rows = []
for key, cnt in calls.items():
    kind, m, n, k, oa, ob = key
    if kind == "gemm":
        A = torch.randn(m, k, device=DEV) if oa == "A k-fast" else torch.randn(k, m, device=DEV)
        B = torch.randn(n, k, device=DEV) if ob == "B k-fast" else torch.randn(k, n, device=DEV)
        sam, sak = (k, 1) if oa == "A k-fast" else (1, m)
        sbk, sbn = (1, k) if ob == "B k-fast" else (n, 1)
        f = lambda: orig(A, sam, sak, B, sbk, sbn, m, n, k, None, "bf16x3")
        byts = 4 * (m * k + n * k + m * n)
    else:
        dy, x = torch.randn(m, n, device=DEV), torch.randn(m, k, device=DEV)
        wb = torch.empty(n * k + n, device=DEV)
        part = torch.empty(lib.hpfg_gemm_tn_splits(m, n, k) * wb.numel(), device=DEV)
        f = lambda: L.check(orig_tn(L.ptr(dy), L.ptr(x), L.ptr(wb), L.ptr(part), m, n, k, 1, torch.cuda.current_stream(DEV).cuda_stream), "tn")
        byts = 4 * (m * n + m * k + n * k)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    rows.append((us * cnt, us, cnt, key, byts, 2.0 * m * n * k))
print(f"{'us/step':>9} {'us':>8} {'calls':>5}  shape (rows, N, K)                                   GB/s     TFLOP/s (fp32-equivalent)")
for tot, us, cnt, key, byts, fl in sorted(rows, reverse=True):
    print(f"{tot:9.1f} {us:8.1f} {cnt:5d}  {str(key):55s} {byts / us / 1e3:7.0f} {fl / us / 1e6:9.1f}")
print("total us/step", sum(r[0] for r in rows))
