#!/bin/bash
# Round-3 GPU call 19: the default bench line of the FINAL code.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 420 python bench.py > gpurun_out/r03_bench_last.log 2>&1; tail -1 gpurun_out/r03_bench_last.log > gpurun_out/r03_bench_line_last.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_line_last.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})
print(json.dumps(d['dp_standin_probe_1gpu']['sweep'])[:300]); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'])"
