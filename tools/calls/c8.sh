mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/c8_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c8_pytest.log
tail -14 gpurun_out/c8_pytest.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err
tail -1 gpurun_out/c8_bench.json | cut -c1-700
RV_FUSE_SWIGLU=0 timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-dp-probe > gpurun_out/c8_bench_unfused.json 2> gpurun_out/c8_bench_unfused.err
tail -1 gpurun_out/c8_bench_unfused.json | cut -c1-300
