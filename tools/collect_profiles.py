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
lines = []
for w in ("sup", "hpfg", "cps", "ctct"):
    fn = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(fn):
        ls = [ln for ln in open(fn) if ln.startswith("{")]
        if ls:
            shutil.copy(fn, os.path.join(dst, f"{prefix}_bench_{w}.json"))
            d = json.loads(ls[-1])
            lines.append(f"{w}: {d['ms_per_step']} ms/step = {d['value']} img/s; step roofline {d['step_roofline']['frac_of_8TBps']} of 8 TB/s; "
                         f"dominant family: {d['roofline']['kernel'][:80]} ({d['roofline']['bound']} {d['roofline']['frac']})")
open(os.path.join(dst, f"{prefix}_other_configs.txt"), "w").write("\n".join(lines) + "\n")
subprocess.run([sys.executable, "tools/family_traffic.py", os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"),
                os.path.join(dst, f"{prefix}_family_traffic.json"), "mt"], check=True)
print("collected", sorted(os.listdir(dst))[-14:])
