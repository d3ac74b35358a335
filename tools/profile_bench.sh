#!/bin/bash
# Kernel-trace profile of one command on the GPU box; writes <tag>_stats.csv under gpurun_out/.
# Usage: tools/profile_bench.sh <tag> <command...>     (run from the repo root, e.g. through gpurun)
set -u
TAG=$1; shift
R=$PWD
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_$TAG" -- "$@" > "$R/gpurun_out/${TAG}_run.log" 2>&1
cd "$R"
DB=$(find "gpurun_out/prof_$TAG" -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" "gpurun_out/${TAG}_stats.csv" > /dev/null
rm -rf "gpurun_out/prof_$TAG"
grep -v '^[WEI]2026' "gpurun_out/${TAG}_run.log" | tail -2
