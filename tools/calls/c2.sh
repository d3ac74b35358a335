set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/c2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c2_pytest.log
tail -4 gpurun_out/c2_pytest.log
timeout 900 bash tools/pmc_hot_kernels.sh > gpurun_out/c2_pmc.log 2>&1
tail -5 gpurun_out/c2_pmc.log
