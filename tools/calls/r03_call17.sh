#!/bin/bash
# Round-3 GPU call 17: GEMM epilogues with v_permlane32_swap (default) vs the ds_bpermute build: micro-benchmark, step A/B with the
# per-kernel table, then the whole GPU test tier on the final library.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for T in _bperm ""; do RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 200 python tools/exp_gemm_lib_ab.py --iters 8 2>&1 | grep "library\|round 1\|checksums"; done | tee gpurun_out/r03_gemm_epilogue_permlane.log
for T in _bperm "" _bperm ""; do
  echo "--- lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r03_line_tmp$T.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r03_line_tmp$T.json')); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'], 'gemm frac', round(d['roofline']['frac'],4), {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
done 2>&1 | tee gpurun_out/r03_step_ab_epilogue_permlane.log
cp gpurun_out/r03_line_tmp.json gpurun_out/r03_bench_line_last.json
( time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) 2>&1 | tee gpurun_out/r03_pytest_gpu_final.log
