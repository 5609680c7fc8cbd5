"""kernel-trace csv -> the last complete forward: per kernel duration and the gap to the previous kernel's end (diagnostics)"""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the last occurrence of pack_weights_kernel starts the last forward
idx = [i for i, r in enumerate(rows) if "pack_weights" in r[2]]
i0, i1 = idx[-2], idx[-1]
prev = None; tot_k = tot_g = 0
for s, e, k in rows[i0:i1]:
    k = re.sub(r"\(anonymous namespace\)::|hpfg_[a-z0-9]*::|void ", "", k).split("(")[0][:70]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{k:70s} dur {(e - s) / 1e3:7.2f} us   gap {gap:6.2f} us")
    tot_k += (e - s) / 1e3; tot_g += gap; prev = e
print(f"sum of durations {tot_k:.1f} us, sum of gaps {tot_g:.1f} us, span {(rows[i1 - 1][1] - rows[i0][0]) / 1e3:.1f} us, period {(rows[i1][0] - rows[i0][0]) / 1e3:.1f} us")
