#!/bin/bash
# Round-3 GPU call 2: A/B experiments - attention block pairing (RV_ATTN_PAIR), GEMM epilogue scheduling (XCD stagger,
# serpentine, sc1 / nt output stores), step-level A/B of the candidates, new parity tests, OmniLMM from pixels.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== attention parity (pair_blocks default on)"
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" 2>&1 | tail -3
echo "=== attention pairing A/B"
for P in 0 1; do echo "RV_ATTN_PAIR=$P"; RV_ATTN_PAIR=$P timeout 300 python tools/bench_hot_kernels.py --iters 8 --only attn; done 2>&1 | tee gpurun_out/r03_attn_pairing_ab.log
echo "=== GEMM epilogue experiments"
( timeout 400 python tools/exp_gemm_epilogue.py --iters 6
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_sc1.so timeout 300 python tools/exp_gemm_epilogue.py --iters 6 --staggers 0,2
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_nt.so timeout 300 python tools/exp_gemm_epilogue.py --iters 6 --staggers 0,2 ) 2>&1 | tee gpurun_out/r03_gemm_epilogue_exp.log
echo "=== baseline bench with the DP stand-in sweep (no CPU contention this time)"
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --dp-probe-wgs 4,8,16,32 > gpurun_out/r03_bench_b.log 2>&1
tail -1 gpurun_out/r03_bench_b.log > gpurun_out/r03_bench_line_b.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_line_b.json')); print(round(d['ms_per_step'],1),'ms', d['roofline']['frac']); print(json.dumps(d['dp_standin_probe_1gpu'])[:1500])"
echo "=== step A/B"
for CFG in "RV_ATTN_PAIR=0" "RV_ATTN_PAIR=1" "RV_GEMM_STAGGER=2" "RV_GEMM_STAGGER=4" "RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_sc1.so"; do
  echo "--- $CFG"
  env $CFG timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s', 'gemm frac', round(d['roofline']['frac'],4), {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
done 2>&1 | tee gpurun_out/r03_step_ab.log
echo "=== LoRA + GQA, tower flag, trainer tests"
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_omnilmm_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "=== full-depth parity from the committed fixtures"
timeout 900 python -m pytest tests/test_zz_baseline_configs_gpu.py -m gpu -x -q -k "full_depth" -s 2>&1 | grep -v "^$" | tail -8
echo "=== OmniLMM from pixels"
timeout 900 python bench.py --omnilmm --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe > gpurun_out/r03_bench_omnilmm_pixels.log 2>&1; tail -1 gpurun_out/r03_bench_omnilmm_pixels.log > gpurun_out/r03_bench_line_omnilmm_pixels.json; tail -c 900 gpurun_out/r03_bench_line_omnilmm_pixels.json
