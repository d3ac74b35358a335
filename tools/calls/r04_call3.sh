#!/bin/bash
# Round-4 GPU call 3: L2 channel camping?  Attention at the bench shape with padded row strides; the DMA-only and MFMA-only
# ablation bodies of the dK/dV kernel under the same padding.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "=== shipped library (RV_ATTN_DKV=4)"
RV_ATTN_DKV=4 timeout 300 python tools/exp_attn_strides.py 0 64 128 192 32 16 0 64
echo "=== shipped library (RV_ATTN_DKV=3)"
RV_ATTN_DKV=3 timeout 300 python tools/exp_attn_strides.py 0 64 0 64
for N in 8 6; do
  echo "=== ablation $N (8 = DMA + barrier only, 6 = MFMA + DMA only)"
  RV_ATTN_DKV=4 RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_abl$N.so timeout 300 python tools/exp_attn_strides.py 0 64 128 0 64
done
} 2>&1 | grep -v "^$" | tee gpurun_out/r04_attn_stride_padding.log
