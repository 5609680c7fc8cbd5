#!/bin/bash
# usage: tools/ab_tree.sh OUT WORKLOAD [reps]: bench.py ms/step of the baseline tree (_base/, a `git archive` of the commit to compare with, built in place)
# against this tree, interleaved (same-box A/B of whole trees: kernels AND host code)
out=$1; wl=$2; reps=${3:-2}
mkdir -p "$(dirname "$out")"
for rep in $(seq $reps); do
  for t in _base .; do
    ms=$(cd $t && python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-f32-line --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$wl tree=$t ms_per_step=$ms" >> "$out"
  done
done
cat "$out"
