#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root): bench line, rocprofv3 kernel trace of the same command,
# the two PMC passes of the HBM traffic (separate runs, counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes),
# the CTCT and HPFG step profiles.  Everything lands under gpurun_out/$1/ ; tools/rocpd_stats.py / step_traffic.py summarise it afterwards.
set -o pipefail
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
# boxes of the pool differ by up to 5 % in every kernel (2.20 - 2.31 ms for the same build): PROFILE_MAX_MS=2.25 stops here on a slow one
if [ -n "$PROFILE_MAX_MS" ]; then
  python3 -c "import json,sys; d=[json.loads(l) for l in open('$OUT/bench.json') if l.startswith('{')][0]; print('ms_per_step', d['ms_per_step']); sys.exit(0 if d['ms_per_step'] <= float('$PROFILE_MAX_MS') else 9)" || exit 9
fi
for w in sup hpfg cps ctct; do python bench.py --workload $w --steps 20 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "bench $w failed"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o mt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-line --no-probe > $OUT/trace.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line > $OUT/pmc_fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line > $OUT/pmc_write.log 2>&1 || exit 4
rocprofv3 --kernel-trace --stats -d $OUT/trace_ctct -o c -- python3 $GRAFT_REPO_ROOT/bench.py --workload ctct --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $OUT/trace_ctct.log 2>&1 || exit 5
rocprofv3 --kernel-trace --stats -d $OUT/trace_hpfg -o h -- python3 $GRAFT_REPO_ROOT/bench.py --workload hpfg --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $OUT/trace_hpfg.log 2>&1 || exit 6
echo profile_round done
