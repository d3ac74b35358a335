mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/c31_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c31_pytest.log
tail -3 gpurun_out/c31_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/c31_bench.log 2>&1; tail -1 gpurun_out/c31_bench.log > gpurun_out/c31_bench_line.json; cut -c1-200 gpurun_out/c31_bench_line.json
timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --no-dp-probe > gpurun_out/c31_lora.log 2>&1; tail -1 gpurun_out/c31_lora.log | cut -c1-200
