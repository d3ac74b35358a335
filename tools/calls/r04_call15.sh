#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r04_lora_dgrad_one_pass.log
: > $L
timeout 900 python -m pytest tests/test_lora_gpu.py -q -x 2>&1 | tail -15 >> $L
timeout 300 python tools/exp_lora_dgrad.py >> $L 2>&1
tail -n 40 $L
