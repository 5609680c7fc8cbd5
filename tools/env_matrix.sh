#!/bin/bash
# A/B of HIP runtime switches on the captured Mean-Teacher step (same box, back to back).  usage: bash tools/env_matrix.sh
run() { echo "$1: $(env $1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-f32-line 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run DEBUG_HIP_GRAPH_BATCH_SIZE=16
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_DYNAMIC_QUEUES=0
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run X=0
