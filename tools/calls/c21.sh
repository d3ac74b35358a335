mkdir -p gpurun_out
timeout 120 python tools/exp_mi16_debug.py 2>&1 | grep -E "rows wrong|deltas"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_lora_gpu.py -m gpu -x -q -k "gemm or swiglu or lora" > gpurun_out/c21_pytest.log 2>&1; tail -3 gpurun_out/c21_pytest.log
RV_GEMM_MI16=1 timeout 300 python tools/bench_hot_kernels.py --iters 10 --only gemm 2>&1 | grep -E "^nn|^tn" > gpurun_out/c21_gemm.log; cat gpurun_out/c21_gemm.log
for v in 0 1; do
  RV_GEMM_MI16=$v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dp-probe > gpurun_out/c21_bench_$v.log 2>&1; echo "MI16=$v: $(tail -1 gpurun_out/c21_bench_$v.log | cut -c1-200)"
done
