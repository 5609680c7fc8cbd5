#!/bin/bash
# Two bench.py ranks on ONE GPU (gloo process group; the kernels, engines and hipIpc windows are the real ones): a rehearsal of the N > 1 paths.
# usage (GPU box, repo root): bash tools/dp2_rehearsal.sh <tag> [bench.py flags]
TAG=$1; shift
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
HPFG_BENCH_ONE_DEVICE=1 HPFG_DP_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-probe "$@" > gpurun_out/dp2_$TAG.json 2> gpurun_out/dp2_$TAG.err
echo "rc=$?"; grep '^{' gpurun_out/dp2_$TAG.json | cut -c1-400; grep -o '"parallelism": "[^"]*"' gpurun_out/dp2_$TAG.json
