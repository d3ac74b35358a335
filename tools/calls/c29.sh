mkdir -p gpurun_out
R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q --durations=3 > gpurun_out/c29_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c29_pytest.log
tail -9 gpurun_out/c29_pytest.log | head -4
timeout 900 python bench.py > gpurun_out/c29_bench.log 2>&1; tail -1 gpurun_out/c29_bench.log > gpurun_out/c29_bench_line.json; cut -c1-260 gpurun_out/c29_bench_line.json
bash tools/profile_bench.sh r02d python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
