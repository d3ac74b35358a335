#!/bin/bash
# HBM traffic (PMC, own passes) of the LoRA-side streaming kernels: bytes per launch / duration
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
L=$R/gpurun_out/r04_lora_skinny_pmc.log
: > $L
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- python $R/tools/exp_lora_skinny.py > /tmp/pmc_$c.log 2>&1
  DB=$(find /tmp/pmc_$c -name '*.db' | head -1)
  echo "== $c" >> $L
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py "$DB" 2>&1 | grep -i "skinny\|dropout\|gemm_tn_256_kernel<EpiStoreF32\|rmsnorm_fwd\|swiglu_fwd" | head -20 >> $L; else tail -5 /tmp/pmc_$c.log >> $L; fi
done
cat $L
