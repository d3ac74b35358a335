#!/bin/bash
# 2-rank REHEARSAL of bench.py's multi-GPU code path (dp_diag, timeline, mode A/B) on the 1-GPU box: gloo backend, both ranks on
# device 0, 2 layers.  Not a measurement - it checks that the first real 8-GPU run cannot die on a Python error.
set -u
export TMPDIR=/tmp RV_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
    bench.py --gpus 2 --steps 2 --warmup 1 --layers 2 --pairs-per-gpu 2 --no-cpu-baseline > gpurun_out/r04_rehearsal_2rank.log 2>&1
echo "exit $?" >> gpurun_out/r04_rehearsal_2rank.log
tail -c 6000 gpurun_out/r04_rehearsal_2rank.log
