mkdir -p gpurun_out
RV_GEMM_NN_W4=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nn" > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c7_pytest.log
tail -5 gpurun_out/c7_pytest.log
for r in 1 2; do
  echo "== A64 (pass $r)"; RV_GEMM_NN_W4=0 timeout 200 python tools/bench_hot_kernels.py --iters 10 --only gemm 2>&1 | grep "^nn"
  echo "== W4 (pass $r)"; RV_GEMM_NN_W4=1 timeout 200 python tools/bench_hot_kernels.py --iters 10 --only gemm 2>&1 | grep "^nn"
done | tee gpurun_out/c7_w4_ab.log
RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_exp.so timeout 200 python tools/exp_gemm_ablate_power.py > gpurun_out/c7_ablate_power.log 2>&1
cat gpurun_out/c7_ablate_power.log | cut -c1-300
