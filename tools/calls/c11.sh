mkdir -p gpurun_out
timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" > gpurun_out/c11_attn.log
cat gpurun_out/c11_attn.log
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c11_pytest.log
tail -12 gpurun_out/c11_pytest.log
