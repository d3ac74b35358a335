mkdir -p gpurun_out
bash tools/probes/mfma_power_probe.sh
timeout 600 python -m pytest tests/test_omnilmm_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "omnilmm or lmhead or resampler or dpo_step" > gpurun_out/c16_pytest.log 2>&1; tail -3 gpurun_out/c16_pytest.log
