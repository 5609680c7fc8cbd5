#!/bin/bash
# usage: tools/r4_ab_lib.sh OUT WORKLOAD lib1.so lib2.so ... : bench.py ms/step per library build, interleaved twice (same-box A/B of builds)
out=$1; wl=$2; shift; shift
mkdir -p "$(dirname "$out")"
for rep in 1 2; do
  for l in "$@"; do
    ms=$(python tools/bench_lib.py $l --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-f32-line --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$wl $l ms_per_step=$ms" >> "$out"
  done
done
cat "$out"
