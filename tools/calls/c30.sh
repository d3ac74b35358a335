mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_omnilmm_gpu.py -m gpu -x -q -k "attn or rope or backward or training_step or dpo_step" > gpurun_out/c30_pytest.log 2>&1; tail -5 gpurun_out/c30_pytest.log
for v in 0 1; do
  RV_FUSE_ROPE_BWD=$v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dp-probe > gpurun_out/c30_bench_$v.log 2>&1; echo "FUSE_ROPE_BWD=$v: $(tail -1 gpurun_out/c30_bench_$v.log | cut -c1-200)"
done
