#!/bin/bash
# Round-4 GPU call 11: L2 prefetch of the epilogue's loads (gate|up of the SwiGLU-backward GEMM, residual of the o / down projections)
# with discard loads a few phases before the end of the K loop - kernel A/B with checksums, parity, step A/B.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for T in _nopf "" _nopf ""; do
  echo "=== lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 300 python tools/exp_gemm_lib_ab.py --iters 10 2>&1 | grep -E "round 1|checksums"
done | tee gpurun_out/r04_gemm_epilogue_l2_prefetch.log
echo "=== parity (default build)"
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or swiglu or linear" 2>&1 | tail -3
echo "=== step"
for T in _nopf "" _nopf ""; do
  echo "--- lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'], 'gemm', round(d['roofline']['frac'],4), {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
done 2>&1 | tee gpurun_out/r04_step_ab_epilogue_l2_prefetch.log
