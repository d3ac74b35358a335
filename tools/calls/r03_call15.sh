#!/bin/bash
# Round-3 GPU call 15: softmax / dS slices under the MFMAs - forward AND dQ kernel (default on) vs the build without.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for T in _nosplit "" _nosplit ""; do
  echo "=== lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn --attn-rounds 3 2>&1 | grep "^attn" | tail -4
done | tee gpurun_out/r03_attn_smsplit_fwd_dq.log
echo "=== parity (default build)"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_omnilmm_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
echo "=== step"
for T in _nosplit "" _nosplit ""; do
  echo "--- lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'])"
done 2>&1 | tee gpurun_out/r03_step_ab_attn_smsplit.log
