#!/bin/bash
# Round-4 state evidence after the LoRA work: whole GPU test tier, smoke, default bench line (with the CPU leg), kernel-trace stats of
# the headline and of the config-5 (LoRA) step, OmniLMM line.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== pytest -m gpu (whole tier)"
( time RV_ROUND=r04 timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r04_pytest_gpu_b.log
echo "=== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
echo "=== default bench"
timeout 1200 python bench.py > gpurun_out/r04_bench_final_b.log 2>&1; tail -1 gpurun_out/r04_bench_final_b.log > gpurun_out/r04_bench_line_final_b.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line_final_b.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:200])"
echo "=== kernel trace stats (headline)"
bash tools/profile_bench.sh r04final_b python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe
head -12 gpurun_out/r04final_b_stats.csv | cut -c1-150
echo "=== LoRA config 5"
timeout 900 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_lora_final.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line_lora_final.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], {k:(round(v['ms_per_step'],1), round(v['frac'],3)) for k,v in d['roofline']['by_kernel'].items()})"
bash tools/profile_bench.sh r04lora python $PWD/bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe
head -24 gpurun_out/r04lora_stats.csv | cut -c1-150
echo "=== OmniLMM from pixels"
timeout 900 python bench.py --omnilmm --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_omnilmm_pixels_b.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_line_omnilmm_pixels_b.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'], d['max_memory_allocated_gb'])"
