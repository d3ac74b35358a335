mkdir -p gpurun_out
R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/c23_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c23_pytest.log
tail -9 gpurun_out/c23_pytest.log | head -8
timeout 900 python bench.py > gpurun_out/c23_bench.log 2>&1; tail -1 gpurun_out/c23_bench.log > gpurun_out/c23_bench_line.json; cut -c1-300 gpurun_out/c23_bench_line.json
bash tools/profile_bench.sh r02c python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe | cut -c1-200
bash tools/collect_pmc_traffic.sh | tail -2
bash tools/pmc_hot_kernels.sh > /dev/null 2>&1
python tools/pmc_table.py gpurun_out > gpurun_out/c23_pmc_table.txt 2>&1; grep -E "^==|MFMA busy" gpurun_out/c23_pmc_table.txt
timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --no-dp-probe > gpurun_out/c23_lora.log 2>&1; tail -1 gpurun_out/c23_lora.log > gpurun_out/c23_lora_line.json; cut -c1-200 gpurun_out/c23_lora_line.json
timeout 600 python bench.py --omnilmm --no-dp-probe > gpurun_out/c23_omni.log 2>&1; tail -1 gpurun_out/c23_omni.log > gpurun_out/c23_omni_line.json; cut -c1-200 gpurun_out/c23_omni_line.json
