"""End of a training step from a rocprofv3 kernel trace (diagnostics): which hardware queue finishes last and what runs in the last 150 us.
usage: python tools/step_tail.py <dir with *_kernel_trace.csv> [n_networks: pack_weights launches per step, default 1]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
npk = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
pk = [i for i, r in enumerate(rows) if "pack_weights" in r["Kernel_Name"]]
a, b = pk[-2 * npk], pk[-npk]          # one whole step (the second to last)
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"{len(step)} kernels, span {(t1 - t0) / 1e3:.1f} us")
qs = {}
for r in step:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = qs.setdefault(q, [s, e, 0, 0]); d[0] = min(d[0], s); d[1] = max(d[1], e); d[2] += e - s; d[3] += 1
for q, (s, e, busy, n) in sorted(qs.items(), key=lambda kv: kv[1][0]):
    print(f"queue {q}: {n:3d} kernels, first start {(s - t0) / 1e3:8.1f} us, last end {(e - t0) / 1e3:8.1f} us, busy {busy / 1e3:8.1f} us")
print("--- kernels ending in the last 150 us:")
for r in sorted(step, key=lambda r: int(r["End_Timestamp"])):
    e = int(r["End_Timestamp"])
    if e > t1 - 150000:
        n = re.sub(r"\(anonymous namespace\)::|hpfg_[a-z0-9]*::|void ", "", r["Kernel_Name"]).split("(")[0][:70]
        print(f"  q{r['Queue_Id']} {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  {n}")
