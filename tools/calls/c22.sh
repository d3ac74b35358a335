mkdir -p gpurun_out
timeout 600 python tools/exp_gemm_variants.py ab > gpurun_out/c22_variants.log 2>&1; tail -20 gpurun_out/c22_variants.log
