mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn or attention" > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c10_pytest.log
tail -8 gpurun_out/c10_pytest.log
for v in 2 3; do
  echo "== RV_ATTN_FWD=$v" >> gpurun_out/c10_attn.log
  RV_ATTN_FWD=$v timeout 200 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" >> gpurun_out/c10_attn.log
done
cat gpurun_out/c10_attn.log
