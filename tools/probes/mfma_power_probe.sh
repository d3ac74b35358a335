#!/bin/bash
# Usage (GPU box, repo root): bash tools/probes/mfma_power_probe.sh  -> gpurun_out/mfma_power_probe.log
mkdir -p gpurun_out
LOG=gpurun_out/mfma_power_probe.log
: > $LOG
for mode in 0 1 2 0 1; do
  ./tools/probes/mfma_power_probe $mode 4 >> $LOG &
  PID=$!
  sleep 2.5
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | sed 's/^/    /' >> $LOG
  wait $PID
done
cat $LOG
