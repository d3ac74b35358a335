#!/bin/bash
# Whole GPU test tier + smoke at the end-of-round code state
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time RV_ROUND=r04 timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 ) 2>&1 | tee gpurun_out/r04_pytest_gpu_final.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 | tee -a gpurun_out/r04_pytest_gpu_final.log
