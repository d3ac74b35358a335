mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn or attention" > gpurun_out/c27_pytest.log 2>&1; tail -4 gpurun_out/c27_pytest.log
timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" > gpurun_out/c27_attn.log
RV_ATTN_DKV=2 timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" | sed 's/^/dkv2: /' >> gpurun_out/c27_attn.log
cat gpurun_out/c27_attn.log
