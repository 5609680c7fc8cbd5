#!/bin/bash
# rocprofv3 kernel stats of a few eager CTCT steps (SegFormer branch diagnostics) -> gpurun_out/ctctp/
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/ctctp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=5 GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o c -- python3 $GRAFT_REPO_ROOT/tools/bench_ctct.py > $OUT/log.txt 2>&1 || exit 5
echo done
