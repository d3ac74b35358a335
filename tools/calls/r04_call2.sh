#!/bin/bash
# Round-4 GPU call 2: where does the dK/dV kernel's time go?  Ablation bodies of version 4 (tools/gen_attn_dkv4.py --ablations),
# per-kernel durations from rocprofv3 kernel traces of the attention micro-benchmark (packed case).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r04_attn_dkv4_ablation.log
: > $OUT
for N in 0 1 2 3 4 5 6 7 8; do
  LIB=$PWD/rlaif-v_amd/librlaifv_hip.so
  [ $N -gt 0 ] && LIB=$PWD/rlaif-v_amd/librlaifv_hip_abl$N.so
  RV_ATTN_DKV=4 RV_HIP_LIB=$LIB bash tools/profile_bench.sh abl$N python $PWD/tools/bench_hot_kernels.py --iters 10 --only attn --attn-case packed --attn-rounds 2 > /dev/null 2>&1
  echo "== dkv4 ablation $N: $(grep -E 'attn_bwd_dkv4|attn_bwd_dq2' gpurun_out/abl${N}_stats.csv | awk -F, '{printf "%s avg %.1f us min %.1f us | ", substr($1,2,22), $4/1000, $5/1000}')" | tee -a $OUT
done
RV_ATTN_DKV=3 bash tools/profile_bench.sh dkv3 python $PWD/tools/bench_hot_kernels.py --iters 10 --only attn --attn-case packed --attn-rounds 2 > /dev/null 2>&1
echo "== dkv3: $(grep -E 'attn_bwd_dkv3|attn_bwd_dq2|attn_fwd2' gpurun_out/dkv3_stats.csv | awk -F, '{printf "%s avg %.1f us min %.1f us | ", substr($1,2,22), $4/1000, $5/1000}')" | tee -a $OUT
