#!/bin/bash
# Round-4 GPU call 12: AdamW fused with the W^T refresh (rv_adamw_step_t) - bit-identity test, step A/B, kernel table of the optimizer part.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== bit identity + optimizer / checkpoint tests"
timeout 900 python -m pytest tests/test_trainer_semantics_gpu.py tests/test_lora_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "=== step A/B"
for F in 0 1 0 1; do
  echo "--- RV_FUSE_ADAMW_T=$F"
  RV_FUSE_ADAMW_T=$F timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'])"
done 2>&1 | tee gpurun_out/r04_step_ab_adamw_transpose.log
echo "=== optimizer kernels"
for F in 0 1; do
  RV_FUSE_ADAMW_T=$F bash tools/profile_bench.sh adamwt$F python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe --no-gemm-timer > /dev/null 2>&1
  echo "--- RV_FUSE_ADAMW_T=$F"; grep -E "adamw|transpose" gpurun_out/adamwt${F}_stats.csv | cut -c1-60,100-200
done | tee -a gpurun_out/r04_step_ab_adamw_transpose.log
