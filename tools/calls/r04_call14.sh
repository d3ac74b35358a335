#!/bin/bash
# A/B: logistic with v_rcp_f32 instead of the IEEE division sequence in the SwiGLU epilogues / kernels
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r04_swiglu_rcp_ab.log
: > $L
for rep in 1 2; do
  for lib in librlaifv_hip_base.so librlaifv_hip.so; do
    echo "== $lib (rep $rep)" >> $L
    RV_HIP_LIB=$PWD/rlaif-v_amd/$lib timeout 300 python tools/exp_gemm_lib_ab.py --iters 10 >> $L 2>&1
  done
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "swiglu or gemm" 2>&1 | tail -5 >> $L
tail -n 60 $L
