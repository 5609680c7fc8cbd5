"""rocprofv3 (ROCm 7.2 default output: a rocpd SQLite database) -> the kernel_stats.csv summary of `--stats` (Name, Calls, TotalDurationNs,
AverageNs, Percentage, MinNs, MaxNs, StdDev), plus launches per step.  usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import math
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), sum(1.0*duration*duration) from kernels group by name").fetchall()
tot = sum(r[2] for r in rows)
out = []
for name, n, s, avg, mn, mx, sq in sorted(rows, key=lambda r: -r[2]):
    var = max(sq / n - avg * avg, 0.0)
    out.append([name, n, int(s), round(avg, 3), round(100.0 * s / tot, 4), int(mn), int(mx), round(math.sqrt(var), 3)])
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
w.writerows(out)
steps = next((r[1] for r in rows if "sgd_kernel" in r[0]), 1)
print(f"# steps={steps} launches/step={sum(r[1] for r in rows) / steps:.1f} kernel time/step={tot / steps / 1e6:.3f} ms", file=sys.stderr)
