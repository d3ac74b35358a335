#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r04_lora_fwd_pre.log
: > $L
timeout 900 python -m pytest tests/test_lora_gpu.py -q -x 2>&1 | tail -5 >> $L
for m in 0 1; do
  RV_LORA_FWD_PRE=$m timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --no-cpu-baseline --no-dp-probe > gpurun_out/r04_lora_bench_fp$m.json 2> gpurun_out/r04_lora_bench_fp$m.err
  python - >> $L <<PY
import json
d=json.loads(open('gpurun_out/r04_lora_bench_fp$m.json').read().strip().splitlines()[-1])
print('RV_LORA_FWD_PRE=$m', d['value'], d['ms_per_step'], d['step_mfma_frac'], d['roofline']['frac'])
for k,v in d['roofline']['by_kernel'].items(): print('   ', k, v['launches'], round(v['avg_launch_ms'],3), round(v['ms_per_step'],1), round(v['frac'],3))
PY
done
cat $L
