#!/bin/bash
# fetched bytes of the upsample-backward kernel under both work mappings (PMC pass: counters + kernel trace only)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/upb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export HPFG_UPB_XCD=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p$v -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line > $OUT/p$v.log 2>&1 || exit 3
  python3 - $OUT/p$v/f_counter_collection.csv <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']=='FETCH_SIZE' and ('upsample_bwd' in r['Kernel_Name']):
        d[r['Grid_Size']].append(float(r['Counter_Value']))
for g,v in d.items(): print('grid',g,'n',len(v),'FETCH_SIZE avg',sum(v)/len(v),'-> MB (x64B... raw KB*2)',sum(v)/len(v)*2/1e3)
PY
done
