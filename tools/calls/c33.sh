mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_omnilmm_gpu.py -m gpu -x -q -s -k "full_width" > gpurun_out/c33.log 2>&1; tail -16 gpurun_out/c33.log
