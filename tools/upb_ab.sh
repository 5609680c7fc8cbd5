#!/bin/bash
# kernel-level A/B of the upsample-backward work mapping (rocprofv3 kernel stats of a short bench run per setting)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/upb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export HPFG_UPB_XCD=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x$v -o u -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-line > $OUT/x$v.log 2>&1 || exit 2
  grep -h "upsample_bwd\|bn_bwd_reduce_pool" $OUT/x$v/*kernel_stats.csv | cut -c1-60,100-400
done
