#!/bin/bash
# Round-4 GPU call 10 (state evidence): whole GPU test tier, smoke, default bench line (with the CPU leg), kernel-trace stats,
# attention PMC, PMC traffic, OmniLMM line.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== pytest -m gpu (whole tier)"
( time RV_ROUND=r04 timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r04_pytest_gpu.log
echo "=== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
echo "=== default bench"
timeout 1200 python bench.py > gpurun_out/r04_bench_final.log 2>&1; tail -1 gpurun_out/r04_bench_final.log > gpurun_out/r04_bench_line_final.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line_final.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})
print(json.dumps(d['dp_standin_probe_1gpu']['sweep'])[:400]); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:200])"
echo "=== kernel trace stats"
bash tools/profile_bench.sh r04final python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe
head -14 gpurun_out/r04final_stats.csv | cut -c1-150
echo "=== attention PMC"
bash tools/pmc_attn_r04.sh r04 2>&1 | tail -5
echo "=== PMC traffic"
bash tools/collect_pmc_traffic.sh 2>&1 | tail -3
echo "=== OmniLMM from pixels"
timeout 900 python bench.py --omnilmm --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_omnilmm_pixels.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_line_omnilmm_pixels.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'], d['max_memory_allocated_gb'])"
