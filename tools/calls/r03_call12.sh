#!/bin/bash
# Round-3 GPU call 12: where does the SwiGLU-backward epilogue's time go?  Ablations (wrong results by construction): no gate|up
# loads / half the output stores, against the shipped kernel on the step's shape.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for T in "" _abl1 _abl2; do
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 200 python tools/exp_gemm_lib_ab.py --iters 8 2>&1 | grep "library\|round 1"
done | tee gpurun_out/r03_swiglu_bwd_epilogue_ablation.log
