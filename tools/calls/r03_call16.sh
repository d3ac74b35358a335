#!/bin/bash
# Round-3 GPU call 16: FINAL state - GPU test tier, smoke, default bench, kernel-trace stats, attention PMC (MFMA busy).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) 2>&1 | tee gpurun_out/r03_pytest_gpu_final.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_bench_final.log 2>&1; tail -1 gpurun_out/r03_bench_final.log > gpurun_out/r03_bench_line_final.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_line_final.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})
print(json.dumps(d['dp_standin_probe_1gpu']['sweep'])[:400]); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'])"
bash tools/profile_bench.sh r03final python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe
head -10 gpurun_out/r03final_stats.csv | cut -c1-130
bash tools/pmc_attn_packed_vs_plain.sh 2>&1 | grep "packed" | cut -c1-330
