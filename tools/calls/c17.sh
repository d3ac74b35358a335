mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nn or swiglu_fused" > gpurun_out/c17_pytest.log 2>&1; tail -5 gpurun_out/c17_pytest.log
for v in 0 1; do
  echo "== RV_GEMM_MI16=$v" >> gpurun_out/c17_gemm.log
  RV_GEMM_MI16=$v timeout 300 python tools/bench_hot_kernels.py --iters 10 --only gemm 2>&1 | grep -E "^nn|^tn" >> gpurun_out/c17_gemm.log
done
cat gpurun_out/c17_gemm.log
