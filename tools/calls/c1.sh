set -x
mkdir -p gpurun_out
(nproc; free -g; lscpu | head -20; rocm-smi --showmeminfo vram | head -8) > gpurun_out/host_info.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1_pytest.log
tail -5 gpurun_out/c1_pytest.log
timeout 300 python tools/bench_hot_kernels.py --iters 5 > gpurun_out/c1_hot.log 2>&1
cat gpurun_out/c1_hot.log
