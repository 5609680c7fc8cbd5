#!/bin/bash
# step-time A/B of the XCD-aware work mappings (wgrad pixel ranges, upsample backward), then their fetched bytes (PMC pass)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/mapab
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-f32-line 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
run HPFG_WGRAD_XCD=1
run HPFG_WGRAD_XCD=0
run HPFG_WGRAD_XCD=1
run HPFG_WGRAD_XCD=0
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export HPFG_WGRAD_XCD=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p$v -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line > $OUT/p$v.log 2>&1 || exit 3
  python3 - $OUT/p$v/f_counter_collection.csv <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']=='FETCH_SIZE':
        k='wgrad' if 'wgrad_bf16x3' in r['Kernel_Name'] else 'upsample_bwd' if 'upsample_bwd' in r['Kernel_Name'] else None
        if k: d[k].append(float(r['Counter_Value']))
for g,v in d.items(): print(g,'launches',len(v),'fetched MB per step', sum(v)*2/1e3/3)
PY
done
