#!/bin/bash
# Round-3 GPU call 7: weight-gradient GEMMs on a low-priority side stream (RV_WGRAD_STREAM) - parity of a training step and
# step-time A/B (no GEMM event timers: overlapped launches make per-launch durations meaningless).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
echo "=== parity with the side stream on"
RV_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_model_parity_gpu.py tests/test_trainer_semantics_gpu.py tests/test_dist_rccl_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "=== step A/B"
for CFG in "RV_WGRAD_STREAM=0" "RV_WGRAD_STREAM=1" "RV_WGRAD_STREAM=1 RV_WGRAD_PRIO=0" "RV_WGRAD_STREAM=0" "RV_WGRAD_STREAM=1"; do
  echo "--- $CFG"
  env $CFG timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'], 'mem', round(d['max_memory_allocated_gb'],1))"
done 2>&1 | tee gpurun_out/r03_step_ab_wgrad_stream.log
