"""Pretty-print a rocprofv3 kernel_stats.csv: per-step time by kernel (steps inferred from the sgd_kernel call count)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = next((int(r["Calls"]) for r in rows if "sgd_kernel" in r["Name"]), 1)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"steps={steps}  total kernel time per step = {tot/steps/1e6:.3f} ms")
groups = {}
for r in rows:
    n = r["Name"]
    short = re.sub(r"\(anonymous namespace\)::|hpfg_\w+::|void ", "", n)
    short = re.sub(r"\(Hpfg.*", "", short)[:70]
    key = short.split("<")[0]
    groups.setdefault(key, [0.0, 0])
    groups[key][0] += float(r["TotalDurationNs"]); groups[key][1] += int(r["Calls"])
for k, (t, c) in sorted(groups.items(), key=lambda x: -x[1][0])[:16]:
    print(f"  {k:40s} {t/steps/1e3:9.1f} us/step  {c/steps:6.1f} calls/step")
print("top kernels:")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    short = re.sub(r"\(anonymous namespace\)::|hpfg_\w+::|void ", "", r["Name"])
    short = re.sub(r"\(Hpfg.*|\(float.*", "", short)[:78]
    print(f"  {short:80s} {float(r['TotalDurationNs'])/steps/1e3:8.1f} us/step  avg {float(r['AverageNs'])/1e3:7.1f} us x {int(r['Calls'])/steps:.1f}")
