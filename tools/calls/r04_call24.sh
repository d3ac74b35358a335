#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== checksums (expect swiglu=c73b528db79d swiglu_bwd=f58cc660a9ae qkv=70dadbd583c0 down+res=967fd15009c6)"
timeout 300 python tools/exp_gemm_lib_ab.py --iters 6 2>&1 | grep -v amdgpu.ids | tail -3
echo "=== full-depth + baseline-config tests"
( RV_ROUND=r04 timeout 1500 python -m pytest tests/test_zz_baseline_configs_gpu.py tests/test_kernels_gpu.py -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) 2>&1 | tee gpurun_out/r04_pytest_gpu_c.log
echo "=== default bench"
timeout 1200 python bench.py > gpurun_out/r04_bench_final_c.log 2>&1; tail -1 gpurun_out/r04_bench_final_c.log > gpurun_out/r04_bench_line_final_c.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line_final_c.json'))
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], 'gemm frac', d['roofline']['frac'], {k:round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
echo "=== kernel trace stats (headline)"
bash tools/profile_bench.sh r04final_c python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dp-probe > /dev/null
head -10 gpurun_out/r04final_c_stats.csv | cut -c1-150
