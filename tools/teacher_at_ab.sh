#!/bin/bash
# usage: tools/teacher_at_ab.sh OUT : Mean-Teacher step with the teacher's forward forked behind layer k of the student's (HPFG_TEACHER_AT), interleaved
out=$1
mkdir -p "$(dirname "$out")"
for rep in 1 2; do
  for k in -1 1 3 5 7 9 13; do
    ms=$(env HPFG_LOSS_ONE=0 HPFG_TEACHER_AT=$k python bench.py --workload mt --steps 50 --warmup 10 --no-cpu-baseline --no-f32-line --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "mt HPFG_TEACHER_AT=$k ms_per_step=$ms" >> "$out"
  done
done
cat "$out"
