mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
bash tools/profile_bench.sh r02b python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe | cut -c1-300
bash tools/profile_bench.sh r02lora python $R/bench.py --lora --seq-len 4096 --pairs-per-gpu 4 --steps 3 --warmup 1 --no-dp-probe | cut -c1-200
cd /tmp
rm -rf /tmp/pmc_l2
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_l2 -- python $R/tools/exp_gemm_l2_reuse.py worker > $R/gpurun_out/c13_l2_worker.log 2>&1
cd $R
python tools/exp_gemm_l2_reuse.py report "$(find /tmp/pmc_l2 -name '*.db' | head -1)" > gpurun_out/c13_l2_reuse.log 2>&1
cat gpurun_out/c13_l2_reuse.log
