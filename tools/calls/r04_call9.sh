#!/bin/bash
# Round-4 GPU call 9: state of the secondary configurations with the round-4 attention backward: LoRA config 5 (L = 4096, 4 pairs),
# OmniLMM from pixels; default bench line with the per-kernel table.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== LoRA config 5"
timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_lora.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line_lora.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'])
for k,v in d['roofline']['by_kernel'].items(): print(' ', k, round(v['ms_per_step'],1), 'ms', round(v['frac'],3))"
echo "=== default bench (no cpu baseline)"
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_a.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line_a.json')); print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'])
print(json.dumps(d['dp_standin_probe_1gpu']['sweep'])[:300])"
