#!/bin/bash
# Round-4 GPU call 8: forward / dQ kernels with the next tile's DMA pieces issued between the S^T MFMAs - parity, timing, stamps.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== parity"
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn or clip" 2>&1 | tail -4
echo "=== micro-benchmark"
timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn --attn-rounds 3 2>&1 | grep "^attn" | tail -4 | tee gpurun_out/r04_attn_dma_interleave.log
echo "=== phase stamps"
RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_aprof.so timeout 200 python tools/exp_attn_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_attn_fwd_dq_phase_profile_after.log
