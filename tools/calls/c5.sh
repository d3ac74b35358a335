mkdir -p gpurun_out
for a in 0 1 2 3 5 6; do
  echo "== dkv3 ablation $a" >> gpurun_out/c5_ablate.log
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_attnexp.so RV_DKV_ABLATE=$a timeout 200 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" >> gpurun_out/c5_ablate.log
done
cat gpurun_out/c5_ablate.log
timeout 600 python tools/exp_gemm_variants.py ab > gpurun_out/c5_gemm_ab.log 2>&1
cat gpurun_out/c5_gemm_ab.log
timeout 200 python tools/exp_gemm_variants.py power > gpurun_out/c5_power.log 2>&1
cat gpurun_out/c5_power.log
