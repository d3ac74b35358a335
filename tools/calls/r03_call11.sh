#!/bin/bash
# Round-3 GPU call 11: PMC passes (MFMA busy, wave-cycle breakdown, traffic) over the GEMM classes incl. the fused-epilogue kernels.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash tools/pmc_hot_kernels.sh gemm > /dev/null 2>&1
python tools/pmc_table.py gpurun_out > gpurun_out/r03_pmc_hot_kernels_gemm.txt 2>&1
cat gpurun_out/r03_pmc_hot_kernels_gemm.txt | grep -E "^==|MFMA busy|WAIT_|ACTIVE_INST_ANY|HBM-side" | head -60
