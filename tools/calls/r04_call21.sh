#!/bin/bash
# half-tile NN GEMM experiment
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r04_gemm_h128.log
: > $L
LIB=$PWD/rlaif-v_amd/librlaifv_hip_h128.so
RV_HIP_LIB=$LIB RV_H128=1 timeout 200 python tools/exp_gemm_h128.py --ldsalloc >> $L 2>&1
for cfg in "0 0" "1 0" "1 600" "1 1100" "0 0" "1 300"; do
  set -- $cfg
  echo "== RV_H128=$1 RV_H128_STAGGER=$2 (ticks per 32-deep phase)" >> $L
  RV_HIP_LIB=$LIB RV_H128=$1 RV_H128_STAGGER=$2 timeout 300 python tools/exp_gemm_lib_ab.py --iters 10 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
