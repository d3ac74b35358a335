#!/bin/bash
# Round-3 GPU call 1: the full-depth parity run (CPU-heavy, ~15 min of the host's 128 cores) in the background; while the
# oracle grinds, the otherwise idle GPU takes the new kernel tests, the bench line with the per-kernel GEMM table and the
# single-GPU DP stand-in sweep, a kernel trace of it, and the hot-kernel micro-benchmark.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/.hip_done
(python tools/full_depth_parity.py > gpurun_out/full_depth_parity.log 2>&1; echo "parity rc=$?" >> gpurun_out/full_depth_parity.log) &
PAR=$!
for i in $(seq 1 120); do [ -f gpurun_out/.hip_done ] && break; sleep 5; done
echo "--- hip part done after $((i*5)) s"; tail -3 gpurun_out/full_depth_parity.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "reduce_copy or swiglu" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_trainer_semantics_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --dp-probe-wgs 4,8,16,32 > gpurun_out/r03_bench_a.log 2>&1
tail -1 gpurun_out/r03_bench_a.log > gpurun_out/r03_bench_line_a.json; tail -c 1500 gpurun_out/r03_bench_line_a.json
timeout 600 python tools/bench_hot_kernels.py --iters 5 > gpurun_out/r03_hot_kernels_a.log 2>&1; cat gpurun_out/r03_hot_kernels_a.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_dp -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --dp-probe-wgs 8 > "$R/gpurun_out/r03_dp_trace_run.log" 2>&1
cd "$R"
DB=$(find /tmp/prof_dp -name '*.db' | head -1)
[ -n "$DB" ] && python tools/dp_overlap_from_trace.py "$DB" gpurun_out/r03_dp_standin_overlap.json
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" gpurun_out/r03_dp_trace_stats.csv > /dev/null
wait $PAR
tail -12 gpurun_out/full_depth_parity.log
ls -la gpurun_out/*.pt
