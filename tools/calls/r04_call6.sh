#!/bin/bash
# Round-4 GPU call 6: dK/dV version 5 with the workspace planes (-delta, -lse / scale read straight into the accumulators),
# 24-bit DMA offsets, nop-free MFMAs: parity (version 3 and 5), timing, phase stamps.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for V in 3 5; do
  echo "=== parity, RV_ATTN_DKV=$V"
  RV_ATTN_DKV=$V timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" 2>&1 | tail -4
done | tee gpurun_out/r04_dkv5_parity.log
echo "=== micro-benchmark"
for V in 3 5 3 5; do
  echo "--- RV_ATTN_DKV=$V"
  RV_ATTN_DKV=$V timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn --attn-rounds 3 2>&1 | grep "^attn" | tail -2
done | tee gpurun_out/r04_attn_dkv5_ab.log
echo "=== phase stamps"
RV_ATTN_DKV=5 RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_prof5v.so timeout 200 python tools/exp_dkv4_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_attn_dkv5_phase_profile.log
