#!/bin/bash
# half-tile NN GEMM experiment, second build (running DMA pointers) + PMC busy numbers for the log
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
L=$R/gpurun_out/r04_gemm_h128_b.log
: > $L
LIB=$R/rlaif-v_amd/librlaifv_hip_h128.so
for cfg in "0 0" "1 0" "0 0" "1 600"; do
  set -- $cfg
  echo "== RV_H128=$1 RV_H128_STAGGER=$2" >> $L
  RV_HIP_LIB=$LIB RV_H128=$1 RV_H128_STAGGER=$2 timeout 300 python tools/exp_gemm_lib_ab.py --iters 10 2>&1 | grep -v amdgpu.ids >> $L
done
cd /tmp
for h in 0 1; do
  rm -rf /tmp/pmc_h$h
  RV_HIP_LIB=$LIB RV_H128=$h timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_h$h -- python $R/tools/exp_gemm_lib_ab.py --iters 2 > /tmp/pmc_h$h.log 2>&1
  DB=$(find /tmp/pmc_h$h -name '*.db' | head -1)
  echo "== PMC RV_H128=$h" >> $L
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py "$DB" gemm_nn 2>&1 | head -40 >> $L; else tail -5 /tmp/pmc_h$h.log >> $L; fi
done
cat $L
