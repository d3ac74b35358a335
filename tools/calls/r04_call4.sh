#!/bin/bash
# Round-4 GPU call 4: phase profile of the dK/dV kernel version 4 from in-kernel s_memtime stamps (full body and ablation bodies).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for N in 0 6 5 3 0; do
  RV_ATTN_DKV=4 RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_prof$N.so timeout 200 python tools/exp_dkv4_prof.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04_attn_dkv4_phase_profile.log
