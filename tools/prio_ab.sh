#!/bin/bash
out=$1; wl=$2
mkdir -p "$(dirname "$out")"
for rep in 1 2 3; do
  for e in "HPFG_PRIO=0" "HPFG_PRIO=1"; do
    ms=$(env HPFG_LOSS_ONE=0 $e python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-f32-line --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$wl $e ms_per_step=$ms" >> "$out"
  done
done
cat "$out"
