mkdir -p gpurun_out
timeout 120 python tools/exp_mi16_debug.py 2>&1 | grep -E "rows wrong|deltas" 
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nn or swiglu_fused" > gpurun_out/c19_pytest.log 2>&1; tail -5 gpurun_out/c19_pytest.log
