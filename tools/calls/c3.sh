set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py -m gpu -x -q -k "attn or golden or full_width or full_size" > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c3_pytest.log
tail -4 gpurun_out/c3_pytest.log
timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn > gpurun_out/c3_hot_v3.log 2>&1
cat gpurun_out/c3_hot_v3.log
RV_ATTN_DKV=2 timeout 300 python tools/bench_hot_kernels.py --iters 10 --only attn > gpurun_out/c3_hot_v2.log 2>&1
cat gpurun_out/c3_hot_v2.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_attn -- python $GRAFT_REPO_ROOT/tools/bench_hot_kernels.py --iters 5 --only attn > $GRAFT_REPO_ROOT/gpurun_out/c3_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/prof_attn -name '*.db' | head -1) gpurun_out/c3_attn_stats.csv | head -8
