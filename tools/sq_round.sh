#!/bin/bash
# SQ instruction / activity counters of every kernel of the eager Mean-Teacher step (separate rocprofv3 --pmc passes, counters + kernel trace
# only) -> gpurun_out/$1/sq_{a,b,c};  tools/sq_summary.py turns them into profiles/<prefix>_step_sq_counters.txt
TAG=${1:-sq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_b -o b -- $B > $OUT/sq_b.log 2>&1 || echo "pass b failed"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq_a -o a -- $B > $OUT/sq_a.log 2>&1 || echo "pass a failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_c -o c -- $B > $OUT/sq_c.log 2>&1 || echo "pass c failed"
echo sq_round done
