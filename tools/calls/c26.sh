mkdir -p gpurun_out
: > gpurun_out/c26_attn.log
for rep in 1 2; do
for v in "" _fp1 _fp2; do
  echo "== lib$v" >> gpurun_out/c26_attn.log
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$v.so timeout 200 python tools/bench_hot_kernels.py --iters 10 --only attn 2>&1 | grep "attn" >> gpurun_out/c26_attn.log
done
done
cat gpurun_out/c26_attn.log
