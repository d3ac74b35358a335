#!/bin/bash
# Round-3 GPU call 10: confirmation of the GPU test tier at the final commit (pass count) + a tile-order (GROUP) sweep of the NN GEMMs.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) 2>&1 | tee gpurun_out/r03_pytest_gpu_final.log
for G in 4 2 8 6; do echo "--- RV_GEMM_GROUP=$G"; RV_GEMM_GROUP=$G timeout 200 python tools/exp_gemm_lib_ab.py --iters 8 2>&1 | grep "round 1"; done | tee gpurun_out/r03_gemm_group_sweep.log
