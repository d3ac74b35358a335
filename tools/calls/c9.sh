mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/c9_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c9_pytest.log
tail -12 gpurun_out/c9_pytest.log
for a in 0 1 2 3 6; do
  echo "== fwd ablation $a" >> gpurun_out/c9_fwd_ablate.log
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_attnexp.so RV_FWD_ABLATE=$a timeout 200 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" >> gpurun_out/c9_fwd_ablate.log
done
cat gpurun_out/c9_fwd_ablate.log
