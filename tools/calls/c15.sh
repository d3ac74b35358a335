mkdir -p gpurun_out
timeout 600 python bench.py --omnilmm --no-dp-probe --steps 3 --warmup 1 > gpurun_out/c15_omni_bench.log 2>&1 || timeout 600 python bench.py --omnilmm --pairs-per-gpu 6 --no-dp-probe --steps 3 --warmup 1 > gpurun_out/c15_omni_bench.log 2>&1
tail -1 gpurun_out/c15_omni_bench.log > gpurun_out/c15_omni_line.json; cut -c1-1500 gpurun_out/c15_omni_line.json
timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/c15_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c15_pytest.log
tail -12 gpurun_out/c15_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
