#!/bin/bash
# Round-3 GPU call 3: GEMM experiment libraries (epilogue prefetch, sc1 / nt stores, persistent tile loop) vs the shipped one -
# speed and bit-identical results; step A/B of the candidates; LoRA config-5 line.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for T in "" _pf _sc1 _nt _pfsc1 _persist _persistpf; do
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 200 python tools/exp_gemm_lib_ab.py --iters 8 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03_gemm_lib_ab.log
echo "=== odd shapes through the persistent loop (ragged M, few tiles)"
for T in "" _persist; do RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 200 python tools/exp_gemm_lib_ab.py --iters 3 --rows 5000 2>&1 | grep "checksums\|library"; done | tee -a gpurun_out/r03_gemm_lib_ab.log
echo "=== step A/B"
for T in "" _pf _persist _persistpf; do
  echo "--- lib$T"
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$T.so timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'], 'gemm frac', round(d['roofline']['frac'],4), {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
done 2>&1 | tee gpurun_out/r03_step_ab_libs.log
echo "=== LoRA config 5 (L = 4096, 4 pairs)"
timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe > gpurun_out/r03_bench_lora.log 2>&1; tail -1 gpurun_out/r03_bench_lora.log > gpurun_out/r03_bench_line_lora.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_line_lora.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
