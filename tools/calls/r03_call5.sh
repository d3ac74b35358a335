#!/bin/bash
# Round-3 GPU call 5: (a) is "packed slower than plain" a clock-ramp artefact? three rounds per process; (b) softmax ILP variant of
# the attention forward (experiment library) - parity, micro-benchmark, step.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for T in "" _ilp; do
  echo "=== lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 300 python tools/bench_hot_kernels.py --iters 8 --only attn --attn-rounds 3 2>&1 | grep "^attn"
done | tee gpurun_out/r03_attn_rounds_softmax_ilp.log
echo "=== parity with the ILP library"
RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_ilp.so timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py -m gpu -x -q -k "attn or golden or forward" 2>&1 | tail -2
echo "=== step A/B"
for T in "" _ilp "" _ilp; do
  echo "--- lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'])"
done 2>&1 | tee gpurun_out/r03_step_ab_softmax_ilp.log
