#!/bin/bash
# Round-3 GPU call 4: attention dispatch experiments (one ranked block per workgroup, XCD map modes) and PMC passes of the
# attention kernels on packed vs plain rows.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo "=== attention parity in single-block mode"
RV_ATTN_PAIR=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" 2>&1 | tail -2
echo "=== attention dispatch A/B"
for CFG in "RV_ATTN_PAIR=1 RV_ATTN_MAP=1" "RV_ATTN_PAIR=2 RV_ATTN_MAP=1" "RV_ATTN_PAIR=1 RV_ATTN_MAP=0" "RV_ATTN_PAIR=1 RV_ATTN_MAP=2" "RV_ATTN_PAIR=2 RV_ATTN_MAP=2" "RV_ATTN_PAIR=2 RV_ATTN_MAP=0"; do
  echo "--- $CFG"; env $CFG timeout 200 python tools/bench_hot_kernels.py --iters 8 --only attn 2>&1 | grep "^attn"
done | tee gpurun_out/r03_attn_dispatch_ab.log
echo "=== PMC packed vs plain"
bash tools/pmc_attn_packed_vs_plain.sh 2>&1 | tail -12
