#!/bin/bash
# PMC passes over the dominant kernels (tools/bench_hot_kernels.py): MFMA busy cycles, LDS bank conflicts, wave-cycle
# breakdown, HBM traffic.  Every pass is its own rocprofv3 run with --kernel-trace only (gpurun refuses --pmc together
# with the sys/hip/hsa trace domains).  Writes gpurun_out/pmc_hot_<pass>.txt.   Usage (repo root, GPU box):
#   tools/pmc_hot_kernels.sh [gemm|attn]
set -u
R=$PWD
ONLY=${1:-}
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$R/gpurun_out/pmc_counters_available.txt" 2>&1 || true
run_pass() {   # name, counters...
  local NAME=$1; shift
  rm -rf "/tmp/pmc_$NAME"
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "/tmp/pmc_$NAME" -- python "$R/tools/bench_hot_kernels.py" --iters 2 \
      ${ONLY:+--only $ONLY} > "$R/gpurun_out/pmc_hot_${NAME}.log" 2>&1
  local DB
  DB=$(find "/tmp/pmc_$NAME" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$R/tools/rocpd_pmc.py" "$DB" > "$R/gpurun_out/pmc_hot_${NAME}.txt" 2>&1; fi
}
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run_pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
cd "$R"
tail -n 40 gpurun_out/pmc_hot_mfma.txt
