#!/bin/bash
# usage: tools/r4_ab.sh OUT WORKLOAD "ENV=a" "ENV=b" ... : bench.py ms/step for each environment setting, interleaved twice (same-box A/B)
out=$1; wl=$2; shift; shift
mkdir -p "$(dirname "$out")"
for rep in 1 2; do
  for e in "$@"; do
    ms=$(env $e python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-f32-line --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$wl $e ms_per_step=$ms" >> "$out"
  done
done
cat "$out"
