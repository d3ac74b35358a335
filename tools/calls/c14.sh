mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_omnilmm_gpu.py -m gpu -x -q -s > gpurun_out/c14_omni.log 2>&1; echo "rc=$?" >> gpurun_out/c14_omni.log
tail -40 gpurun_out/c14_omni.log
