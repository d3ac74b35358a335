#!/bin/bash
# Round-3 GPU call 6: milestone - the whole GPU test tier, the default bench line, kernel-trace stats and the PMC traffic passes.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== pytest -m gpu (whole tier)"
( time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r03_pytest_gpu.log
echo "=== default bench"
timeout 900 python bench.py > gpurun_out/r03_bench_c.log 2>&1; tail -1 gpurun_out/r03_bench_c.log > gpurun_out/r03_bench_line_c.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_line_c.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})
print(json.dumps(d['dp_standin_probe_1gpu'])[:700]); print(json.dumps(d['cpu_baseline'])[:900])"
echo "=== kernel trace stats"
bash tools/profile_bench.sh r03c python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe
head -14 gpurun_out/r03c_stats.csv | cut -c1-150
echo "=== PMC traffic"
bash tools/collect_pmc_traffic.sh 2>&1 | tail -5
