#!/bin/bash
# Round-4 GPU call 7: dK/dV version 5 as the default - whole GPU test tier (incl. the new full-depth cases: config-2 packed step,
# conditioned variants, step-0 self-consistency, LoRA vs the reference on merged weights), step A/B version 3 vs 5.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== pytest -m gpu (whole tier)"
( time RV_ROUND=r04 timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/r04_pytest_gpu.log
echo "=== step A/B"
for V in 3 5 3 5; do
  echo "--- RV_ATTN_DKV=$V"
  RV_ATTN_DKV=$V timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'])"
done 2>&1 | tee gpurun_out/r04_step_ab_dkv5.log
