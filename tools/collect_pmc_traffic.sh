#!/bin/bash
# HBM traffic per kernel launch from PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one
# bench step, kernel trace only (MI355X_MICROARCH.md HBM section).  Writes gpurun_out/pmc_hbm_traffic.json.
# Usage (repo root, on the GPU box): tools/collect_pmc_traffic.sh
set -u
R=$PWD
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/pmc_$C"
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d "/tmp/pmc_$C" -- python "$R/bench.py" --steps 1 --warmup 1 \
      --no-cpu-baseline --no-gemm-timer --no-dp-probe > "$R/gpurun_out/pmc_$C.log" 2>&1
done
cd "$R"
python tools/rocpd_pmc.py --json "$(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1)" \
    > gpurun_out/pmc_hbm_traffic.json
tail -c 600 gpurun_out/pmc_hbm_traffic.json
