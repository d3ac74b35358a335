mkdir -p gpurun_out
: > gpurun_out/c28_attn.log
for rep in 1 2 3; do
for v in "" _s1; do
  echo "== lib$v" >> gpurun_out/c28_attn.log
  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip$v.so timeout 200 python tools/bench_hot_kernels.py --iters 20 --only attn 2>&1 | grep "attn packed" >> gpurun_out/c28_attn.log
done
done
cat gpurun_out/c28_attn.log
