"""HBM traffic of one training step from two rocprofv3 PMC passes (diagnostics for DESIGN.md section 6).

usage: python tools/step_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>
Counters are in KiB per dispatch.  FETCH_SIZE is doubled, as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on
gfx950 (every kernel here reads float4 / b128); WRITE_SIZE is taken as is.  The last whole step (delimited by ema_kernel) is summed per
kernel family and set against the algorithmic 4.73 GB of SURVEY.md section 8(d).
"""
import csv
import glob
import re
import sys


def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "ema_kernel" in r["Kernel_Name"]]
    return rows[idx[-2] + 1: idx[-1] + 1]


def fam(n):
    if "fused_bwd" in n:
        return "fused dgrad + wgrad (thin layers)"
    if "wgrad_bf16x3" in n:
        return "wgrad"
    if "conv1x1" in n:
        return "conv 1x1 (fwd + dgrad)"
    if "conv_thin" in n:
        return "conv 3x3 forward"
    if "conv_bf16x3" in n:
        return "conv 3x3 dgrad" if re.search(r">, (0|4), ", n) else "conv 3x3 forward"
    if "conv_first" in n:
        return "first conv"
    if "bn_" in n:
        return "batchnorm reduce / finalize"
    if "slab_reduce" in n:
        return "slab reduce"
    if "pool" in n or "upsample" in n:
        return "pool / upsample routing"
    if "loss" in n or "channel_sum" in n:
        return "loss + bias sums"
    if "pack" in n or "sgd" in n or "ema" in n:
        return "weights: pack, SGD, EMA"
    return "rest"


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
assert len(fe) == len(wr), (len(fe), len(wr))
acc = {}
for a, b in zip(fe, wr):
    assert a["Kernel_Name"] == b["Kernel_Name"]
    k = fam(a["Kernel_Name"])
    e = acc.setdefault(k, [0.0, 0.0, 0])
    e[0] += 2 * float(a["Counter_Value"]) * 1024
    e[1] += float(b["Counter_Value"]) * 1024
    e[2] += 1
tr = sum(v[0] for v in acc.values())
tw = sum(v[1] for v in acc.values())
print(f"{len(fe)} launches in the step; read {tr / 1e9:.3f} GB, written {tw / 1e9:.3f} GB, total {(tr + tw) / 1e9:.3f} GB  (algorithmic: 4.731 GB)")
for k, v in sorted(acc.items(), key=lambda x: -(x[1][0] + x[1][1])):
    print(f"  {k:30s} {v[2]:4d} launches  read {v[0] / 1e6:8.1f} MB  written {v[1] / 1e6:8.1f} MB")
