mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/c12_bench.log 2>&1; tail -1 gpurun_out/c12_bench.log > gpurun_out/c12_bench_line.json
cut -c1-400 gpurun_out/c12_bench_line.json
bash tools/profile_bench.sh r02b python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe | cut -c1-300
bash tools/collect_pmc_traffic.sh | tail -3
bash tools/pmc_hot_kernels.sh > /dev/null 2>&1
python tools/pmc_table.py gpurun_out > gpurun_out/c12_pmc_table.txt 2>&1; tail -25 gpurun_out/c12_pmc_table.txt
timeout 600 python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --no-dp-probe > gpurun_out/c12_lora.log 2>&1; tail -1 gpurun_out/c12_lora.log > gpurun_out/c12_lora_line.json
cut -c1-400 gpurun_out/c12_lora_line.json
bash tools/profile_bench.sh r02lora python bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --steps 3 --warmup 1 --no-dp-probe | cut -c1-200
