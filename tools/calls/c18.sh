mkdir -p gpurun_out
timeout 120 python tools/exp_mi16_debug.py > gpurun_out/c18_dbg.log 2>&1; head -70 gpurun_out/c18_dbg.log
