"""Per-kernel SQ counters of one eager Mean-Teacher step from the passes of tools/sq_round.sh (diagnostics / profiles).
usage: python tools/sq_summary.py gpurun_out/<tag> > profiles/<prefix>_step_sq_counters.txt
One line per kernel instantiation: launches per step, then per-LAUNCH averages of the dynamic instruction counts (all waves), the MFMA share of the
issued vector instructions, LDS bank-conflict cycles per LDS-active cycle, and -- where the pass worked -- the busy fractions."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for sub in ("sq_a", "sq_b", "sq_c"):
    fs = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|hpfg_[a-z0-9]*::|void ", "", n)
    return n.split("(")[0][:86]


rows = []
for k, d in acc.items():
    per = {c: v / max(1, len(cnt[k][c])) for c, v in d.items()}
    n = max(len(s) for s in cnt[k].values())
    rows.append((per.get("SQ_INSTS_VALU", 0) * n, k, n, per))
print("# bench.py --no-graph --steps 2 --warmup 1 under rocprofv3 --pmc (three passes); per-launch averages over all waves of the launch")
print("# kernel | launches in the run | VALU SALU MFMA LDS VMEM_RD VMEM_WR instructions | VALU per MFMA | LDS conflict cycles / LDS active cycles | "
      "issue-active share of wave cycles | MFMA-busy share of the launch's SIMD-cycles | launch length under the profiler")
for _, k, n, p in sorted(rows, reverse=True):
    g = lambda c: p.get(c, float("nan"))
    mf = g("SQ_INSTS_MFMA")
    line = (f"{short(k):86s} n={n:4d}  VALU {g('SQ_INSTS_VALU'):11.4g} SALU {g('SQ_INSTS_SALU'):11.4g} MFMA {mf:11.4g} LDS {g('SQ_INSTS_LDS'):11.4g} "
            f"RD {g('SQ_INSTS_VMEM_RD'):10.4g} WR {g('SQ_INSTS_VMEM_WR'):10.4g} | VALU/MFMA {g('SQ_INSTS_VALU') / mf if mf and mf == mf else float('nan'):7.2f} "
            f"| conflict/active {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE') if g('SQ_LDS_IDX_ACTIVE') else float('nan'):5.2f} "
            f"| active/wave-cycles {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') == g('SQ_WAVE_CYCLES') and g('SQ_WAVE_CYCLES') else float('nan'):5.2f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in p and p.get("GRBM_GUI_ACTIVE"):
        # SQ_VALU_MFMA_BUSY_CYCLES: SIMD-cycles with the matrix pipe busy, summed over the 1024 SIMDs (= 16 x the MFMA count here);
        # GRBM_GUI_ACTIVE: busy GPU clocks, one instance per XCD (8 summed) -> the launch offers GUI_ACTIVE / 8 x 1024 SIMD-cycles
        line += f" | MFMA busy {p['SQ_VALU_MFMA_BUSY_CYCLES'] / (p['GRBM_GUI_ACTIVE'] * 128.0):5.3f} | {p['GRBM_GUI_ACTIVE'] / 8 / 2.4e3:7.1f} us at 2.4 GHz"
    print(line)
