"""HBM bytes per launch of the conv kernel families of one Mean-Teacher step, from the two rocprofv3 PMC passes of tools/profile_round.sh
(FETCH_SIZE doubled per MI355X_MICROARCH.md for 16-byte-per-lane streaming reads; WRITE_SIZE as it is) -> profiles/<prefix>_family_traffic.json,
which bench.py reads for `roofline.traffic`.   usage: python tools/family_traffic.py <fetch pass dir> <write pass dir> <out.json> [workload]

Families follow bench.py's (pass x layer class); the PMC rows only carry kernel names, so the mapping is by kernel: the chunked split-bf16 conv
kernel (conv_bf16x3_kernel) runs exactly the channel-rich 3x3 layers -- source kinds 1 / 2 / 3 (BatchNorm+LeakyReLU, max-pooled, concat) are
forward launches, kind 4 (dZ) input gradients; conv_thin_kernel runs the thin forward layers (and, with a dZ source, one channel-rich input
gradient); fused_bwd_kernel the fused dgrad + wgrad passes; wgrad_bf16x3_kernel<.., 9> the separate 3x3 weight gradients."""
import csv
import glob
import json
import re
import sys


def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "ema_kernel" in r["Kernel_Name"]]
    return rows[idx[-2] + 1: idx[-1] + 1]


def fam(n):
    if "conv_bf16x3_kernel" in n:
        m = re.search(r">, (\d), (true|false|\d)>", n)      # (kind, backward-epilogue variant)
        kind = int(m.group(1)) if m else -1
        if kind in (1, 2, 3):
            return "forward conv, channel-rich 3x3 (>= 32 channels in and out)"
        if kind == 4:
            return "input gradient (separate dgrad), channel-rich 3x3 (>= 32 channels in and out)"
    if "conv_thin_kernel" in n:
        return "forward conv, thin 3x3 (< 32 channels in or out: the 224 x 224 / 112 x 112 layers and out_conv)"
    if "wgrad_bf16x3_kernel" in n and re.search(r", 9>", n):
        return "weight gradient (separate wgrad), channel-rich 3x3 (>= 32 channels in and out)"
    if "fused_bwd_kernel" in n:
        return "fused dgrad + wgrad (thin and 32-channel layers)"
    return None


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
assert len(fe) == len(wr), (len(fe), len(wr))
acc = {}
for a, b in zip(fe, wr):
    assert a["Kernel_Name"] == b["Kernel_Name"]
    k = fam(a["Kernel_Name"])
    if k is None:
        continue
    e = acc.setdefault(k, [0.0, 0])
    e[0] += 2 * float(a["Counter_Value"]) * 1024 + float(b["Counter_Value"]) * 1024
    e[1] += 1
out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "mt",
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-graph --steps 2` (tools/profile_round.sh); FETCH_SIZE x 2 per MI355X_MICROARCH.md",
       "families": {k: {"launches_per_step": v[1], "hbm_bytes_per_step": int(v[0]), "hbm_bytes_per_launch": int(v[0] / v[1])} for k, v in acc.items()}}
import subprocess  # noqa: E402
out["command"] = "bench.py --no-graph --steps 2 --warmup 1 (under rocprofv3 --pmc, tools/profile_round.sh)"
try:
    out["commit"] = "commit " + subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
except Exception:
    out["commit"] = "commit n/a"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
