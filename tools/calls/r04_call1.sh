#!/bin/bash
# Round-4 GPU call 1: dK/dV kernel version 4 (two sub-tiles in flight, 3-stage ring) - parity, micro-benchmark A/B vs version 3,
# step A/B; LoRA pinned against the reference on merged weights (tiny fixtures); host input pipeline on the GPU box's cores.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== parity, RV_ATTN_DKV=4"
RV_ATTN_DKV=4 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" 2>&1 | tail -15 | tee gpurun_out/r04_dkv4_parity.log
echo "=== micro-benchmark"
for V in 3 4 3 4; do
  echo "--- RV_ATTN_DKV=$V"
  RV_ATTN_DKV=$V timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn --attn-rounds 3 2>&1 | grep "^attn" | tail -4
done | tee gpurun_out/r04_attn_dkv4_ab.log
echo "=== step A/B"
for V in 3 4 3 4; do
  echo "--- RV_ATTN_DKV=$V"
  RV_ATTN_DKV=$V timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dp-probe --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1),'ms', round(d['value'],3),'pairs/s loss', d['loss'])"
done 2>&1 | tee gpurun_out/r04_step_ab_dkv4.log
echo "=== LoRA vs the reference on merged weights (tiny)"
timeout 600 python -m pytest tests/test_lora_gpu.py -m gpu -x -q -k "merged and tiny" 2>&1 | tail -8 | tee gpurun_out/r04_lora_merged_tiny.log
echo "=== host pipeline"
timeout 600 python tools/host_pipeline_bench.py --pairs 256 --workers 1,4,8,16 --out gpurun_out/r04_host_pipeline_gpubox.json 2>&1 | grep -v WARNING | tail -12
