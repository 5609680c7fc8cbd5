"""Summaries of a tools/profile_round.sh run (gpurun_out/<tag>/) -> profiles/<prefix>_* (tracked).  usage: python tools/collect_profiles.py r02b r02"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
dst = "profiles"


def stats(db, out):
    subprocess.run([sys.executable, "tools/rocpd_stats.py", db, out], check=True)


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{prefix}_bench.json"))
stats(glob.glob(os.path.join(src, "trace", "*.db"))[0], os.path.join(dst, f"{prefix}_mt_step_kernel_stats.csv"))
stats(glob.glob(os.path.join(src, "trace_ctct", "*.db"))[0], os.path.join(dst, f"{prefix}_ctct_step_kernel_stats.csv"))
stats(glob.glob(os.path.join(src, "trace_hpfg", "*.db"))[0], os.path.join(dst, f"{prefix}_hpfg_step_kernel_stats.csv"))
for which in ("fetch", "write"):
    f = glob.glob(os.path.join(src, f"pmc_{which}", "*counter_collection.csv"))[0]
    shutil.copy(f, os.path.join(dst, f"{prefix}_step_pmc_{'FETCH' if which == 'fetch' else 'WRITE'}_SIZE.csv"))
with open(os.path.join(dst, f"{prefix}_step_traffic.txt"), "w") as f:
    f.write(f"# python tools/step_traffic.py gpurun_out/{tag}/pmc_fetch gpurun_out/{tag}/pmc_write   (bench.py --no-graph --steps 2; FETCH_SIZE doubled per MI355X_MICROARCH.md)\n")
    f.write(subprocess.run([sys.executable, "tools/step_traffic.py", os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write")], check=True,
                           capture_output=True, text=True).stdout)
shutil.copy(os.path.join(src, "other_configs.txt"), os.path.join(dst, f"{prefix}_other_configs.txt"))
line = [ln for ln in open(os.path.join(src, "ctct.txt")) if ln.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, f"{prefix}_ctct_bench.json"), "w"), indent=1)
print("collected", sorted(os.listdir(dst))[-12:])
