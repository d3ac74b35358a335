#!/bin/bash
# Round-3 GPU call 9 (final state): whole GPU test tier, default bench line, kernel-trace stats, PMC traffic, LoRA and OmniLMM lines.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== pytest -m gpu (whole tier)"
( time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r03_pytest_gpu_final.log
echo "=== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
echo "=== default bench"
timeout 900 python bench.py > gpurun_out/r03_bench_final.log 2>&1; tail -1 gpurun_out/r03_bench_final.log > gpurun_out/r03_bench_line_final.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_line_final.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})
print(json.dumps(d['dp_standin_probe_1gpu']['sweep'])[:500]); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'], d['cpu_baseline'].get('full_depth_measured',{}).get('value'))"
echo "=== kernel trace stats"
bash tools/profile_bench.sh r03final python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe
head -12 gpurun_out/r03final_stats.csv | cut -c1-140
echo "=== PMC traffic"
bash tools/collect_pmc_traffic.sh 2>&1 | tail -3
echo "=== LoRA config 5"
timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r03_bench_line_lora_final.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_line_lora_final.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'])"
echo "=== OmniLMM from pixels"
timeout 900 python bench.py --omnilmm --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r03_bench_line_omnilmm_pixels_final.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_line_omnilmm_pixels_final.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'], d['max_memory_allocated_gb'])"
