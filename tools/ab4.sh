#!/bin/bash
# usage: tools/ab4.sh OUT WORKLOAD "ENV=a" "ENV=b" ... : like r4_ab.sh with four alternations (differences below 0.5 %)
out=$1; wl=$2; shift; shift
mkdir -p "$(dirname "$out")"
for rep in 1 2 3 4; do
  for e in "$@"; do
    ms=$(env $e python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-f32-line --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$wl $e ms_per_step=$ms" >> "$out"
  done
done
cat "$out"
