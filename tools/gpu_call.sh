#!/bin/bash
# One parametrised GPU-box call (replaces the 60 one-off tools/calls/*.sh scripts of rounds 2-4):
#   gpurun --timeout 900 -- 'bash tools/gpu_call.sh <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<round>/ (merged back by gpurun); copy what should be judged into profiles/.
# Stages:
#   attn          attention kernel tests (incl. L = 4096 rows, pad-free rows)
#   kernels       tests/test_kernels_gpu.py
#   padfree       model-level pad-free tests
#   cfg45         full-depth config 4 / 5 parity (tests/test_zzz_configs45_full_depth_gpu.py)
#   gputests      the whole GPU tier (pytest -m gpu), as the driver runs it
#   yardstick     tools/yardstick_hipblaslt.py
#   bench         default bench.py line (headline config)
#   benchfast     headline config without the CPU baseline / DP probe
#   ragged        bench.py --ragged, pad-free vs rectangular
#   lora          bench.py --lora --seq-len 4096 --pairs-per-gpu 4
#   omnilmm       bench.py --omnilmm (config 4 from pixels)
#   prof          rocprofv3 --kernel-trace --stats of the headline bench (3 steps)
#   pmc           HBM traffic per GEMM launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
#   pmcattn       PMC passes of the attention kernels at the bench shape (matrix-pipe busy, wait / issue-stall / active fractions)
#   fwd3          forward attention version 3 against version 2: correctness on 19 shapes, 30-launch stress, timing A/B
#   smoke         __graft_entry__.smoke()
R=${RV_ROUND:-r06}
export RV_ROUND=$R
OUT=gpurun_out/$R
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for st in "$@"; do
  echo "=== stage $st $(date +%T)"
  case $st in
    attn)      timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn" 2>&1 | tail -15 | tee $OUT/pytest_attn.log ;;
    kernels)   timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -15 | tee $OUT/pytest_kernels.log ;;
    padfree)   timeout 900 python -m pytest tests/test_model_parity_gpu.py -q -x -s -k "pad_free or gradient_checkpointing or matches_reference_golden" 2>&1 | tail -25 | tee $OUT/pytest_padfree.log ;;
    cfg45)     timeout 2400 python -m pytest tests/test_zzz_configs45_full_depth_gpu.py -q -s 2>&1 | tail -60 | tee $OUT/pytest_cfg45.log ;;
    gputests)  timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 | tee $OUT/pytest_gpu.log ;;
    yardstick) timeout 600 python tools/yardstick_hipblaslt.py 2>&1 | tee $OUT/yardstick_hipblaslt.log | tail -40 ;;
    bench)     timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log; tail -c 1500 $OUT/bench_line.json ;;
    benchfast) timeout 600 python bench.py --no-cpu-baseline --no-dp-probe > $OUT/bench_line_fast.json 2> $OUT/bench_fast_err.log; head -c 600 $OUT/bench_line_fast.json; echo ;;
    ragged)    for pf in 1 0; do RV_PAD_FREE=$pf timeout 600 python bench.py --ragged --pairs-per-gpu ${RV_RAGGED_PAIRS:-12} --no-dp-probe --steps 4 > $OUT/bench_line_ragged_padfree$pf.json 2> $OUT/bench_ragged_err$pf.log; head -c 500 $OUT/bench_line_ragged_padfree$pf.json; echo; done ;;
    lora)      timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --no-dp-probe > $OUT/bench_line_lora.json 2> $OUT/bench_lora_err.log; head -c 600 $OUT/bench_line_lora.json; echo ;;
    omnilmm)   timeout 900 python bench.py --omnilmm --no-dp-probe > $OUT/bench_line_omnilmm_pixels.json 2> $OUT/bench_omnilmm_err.log; head -c 600 $OUT/bench_line_omnilmm_pixels.json; echo ;;
    prof)      bash tools/profile_bench.sh $R/bench_kernel python $PWD/bench.py --steps 3 --no-cpu-baseline --no-dp-probe --no-gemm-timer; head -25 gpurun_out/$R/bench_kernel_stats.csv ;;
    pmc)       bash tools/collect_pmc_traffic.sh > $OUT/pmc_collect.log 2>&1; cp gpurun_out/pmc_hbm_traffic.json $OUT/pmc_hbm_traffic.json; tail -c 400 $OUT/pmc_hbm_traffic.json; echo ;;
    pmcattn)   bash tools/pmc_attn_r04.sh $R 2>&1 | tail -4; cp gpurun_out/pmc_attn_$R.txt $OUT/pmc_attn.txt ;;
    fwd3)      timeout 600 python tools/exp_attn_fwd3.py 2>&1 | tee $OUT/attn_fwd3.log | tail -16; timeout 300 python tools/exp_attn_fwd3.py --stress 2>&1 | tail -1 ;;
    smoke)     timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log ;;
    *)         echo "unknown stage $st" ;;
  esac
done
echo "=== done $(date +%T)"
