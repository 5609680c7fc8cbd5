"""Per-queue timeline of one training step from a rocprofv3 kernel trace (diagnostics).

usage: python tools/step_timeline.py <dir with *_kernel_trace.csv> [--full]
Prints, for the second-to-last step (delimited by ema_kernel), when each hardware queue starts / ends and how long both were busy.
"""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"Cfg<([^>]*)>, (\d)", n)
    if m:
        return ("conv1x1 " if "1x1" in n else "conv ") + m.group(1).replace(" ", "") + " k" + m.group(2)
    m = re.search(r"wgrad_bf16x3_kernel<([^>]*)>", n)
    if m:
        return "wgrad " + m.group(1).replace(" ", "")
    return n.split("(")[0][-40:]


f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ema_kernel" in r["Kernel_Name"]]
a, b = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
step = rows[a:b]
print(len(step), "kernels; span", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, "us")
qs = {}
for r in step:
    qs.setdefault(r["Queue_Id"], []).append(r)
for q, rs in qs.items():
    s, e = int(rs[0]["Start_Timestamp"]) - t0, int(rs[-1]["End_Timestamp"]) - t0
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}: {len(rs)} kernels, first start {s / 1e3:.1f} us, last end {e / 1e3:.1f} us, busy {busy / 1e3:.1f} us; first: {short(rs[0]['Kernel_Name'])}")
if "--full" in sys.argv:
    for r in step:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:6.1f}  q{r['Queue_Id']} {short(r['Kernel_Name'])}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
