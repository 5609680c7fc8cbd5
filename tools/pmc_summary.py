"""rocprofv3 --pmc ... --output-format csv -> one line per kernel name: counter values of its LAST dispatch (diagnostics).
usage: python tools/pmc_summary.py <dir> [name filter]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
last = {}
for r in csv.DictReader(open(f)):
    if flt and flt not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"]
    d = last.setdefault(k, {})
    did = int(r["Dispatch_Id"])
    if d.get("_id", -1) < did:
        d.clear()
        d["_id"] = did
    if d["_id"] == did:
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in last.items():
    name = re.sub(r"\(anonymous namespace\)::|hpfg_[a-z0-9]*::", "", k).split("(")[0][:70]
    print(name, " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items()) if c != "_id"))
