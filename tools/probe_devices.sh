#!/bin/bash
# does a 1-GPU lease expose more than one LOGICAL device (CPX / DPX compute partitions)?  If so the 2-rank RCCL
# test can run there; if not, one line and stop.
OUT=gpurun_out/${RV_ROUND:-r06}
mkdir -p $OUT
{
  echo "== rocm-smi --showcomputepartition"; rocm-smi --showcomputepartition 2>&1 | head -20
  echo "== rocm-smi --showmemorypartition"; rocm-smi --showmemorypartition 2>&1 | head -12
  echo "== torch.cuda.device_count()"; python -c "import torch; print(torch.cuda.device_count(), [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())])"
  echo "== rocminfo agents"; rocminfo 2>/dev/null | grep -c "Device Type:.*GPU"
} 2>&1 | tee $OUT/device_partition_probe.log
