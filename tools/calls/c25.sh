mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_omnilmm_gpu.py -m gpu -x -q -s -k "eva" > gpurun_out/c25.log 2>&1; tail -15 gpurun_out/c25.log
