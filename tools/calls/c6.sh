mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c6_pytest.log
tail -14 gpurun_out/c6_pytest.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/c6_bench.json 2> gpurun_out/c6_bench.err
cat gpurun_out/c6_bench.json
bash tools/profile_bench.sh r02a python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dp-probe
head -24 gpurun_out/r02a_stats.csv
