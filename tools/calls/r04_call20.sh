#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r04_nt_skinny.log
: > $L
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_nt" 2>&1 | tail -8 >> $L
timeout 600 python -m pytest tests/test_lora_gpu.py -q -x 2>&1 | tail -4 >> $L
for m in 0 1; do echo "== RV_GEMM_NT_SKINNY=$m" >> $L; RV_GEMM_NT_SKINNY=$m timeout 300 python tools/exp_lora_skinny.py 2>&1 | grep -v amdgpu.ids | grep "t = xd\|dt\|sum" >> $L; done
cat $L
