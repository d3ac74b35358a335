mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_lora_gpu.py -m gpu -x -q -k "gemm or swiglu or lora" > gpurun_out/c20_pytest.log 2>&1; tail -4 gpurun_out/c20_pytest.log
for v in 0 1; do
  echo "== RV_GEMM_MI16=$v" >> gpurun_out/c20_gemm.log
  RV_GEMM_MI16=$v timeout 300 python tools/bench_hot_kernels.py --iters 10 --only gemm 2>&1 | grep -E "^nn|^tn" >> gpurun_out/c20_gemm.log
done
cat gpurun_out/c20_gemm.log
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dp-probe > gpurun_out/c20_bench.log 2>&1; tail -1 gpurun_out/c20_bench.log | cut -c1-330
