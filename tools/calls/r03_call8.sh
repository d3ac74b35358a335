#!/bin/bash
# Round-3 GPU call 8: tail split of the weight-gradient GEMMs - correctness, micro-benchmark per penalty, step A/B.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_tn" 2>&1 | tail -2
( RV_TN_TAIL_SPLIT=0 timeout 200 python tools/exp_tn_tail.py
  RV_TN_TAIL_SPLIT=1 RV_TN_TAIL_PENALTY=0.0 timeout 200 python tools/exp_tn_tail.py
  RV_TN_TAIL_SPLIT=1 RV_TN_TAIL_PENALTY=0.02 timeout 200 python tools/exp_tn_tail.py
  RV_TN_TAIL_SPLIT=1 RV_TN_TAIL_PENALTY=0.04 timeout 200 python tools/exp_tn_tail.py ) 2>&1 | grep -v amdgpu | tee gpurun_out/r03_tn_tail_split.log
echo "=== step A/B"
for CFG in "RV_TN_TAIL_SPLIT=0" "RV_TN_TAIL_SPLIT=1 RV_TN_TAIL_PENALTY=0.02" "RV_TN_TAIL_SPLIT=1 RV_TN_TAIL_PENALTY=0.04" "RV_TN_TAIL_SPLIT=0" "RV_TN_TAIL_SPLIT=1 RV_TN_TAIL_PENALTY=0.02"; do
  echo "--- $CFG"
  env $CFG timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'], 'tn', round(d['roofline']['by_kernel']['tn']['frac'],3), round(d['roofline']['by_kernel']['tn']['ms_per_step'],1))"
done 2>&1 | tee gpurun_out/r03_step_ab_tn_tail.log
