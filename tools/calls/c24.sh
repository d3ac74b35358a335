mkdir -p gpurun_out
for i in 1 2; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c24_bench_$i.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/c24_bench_$i.log").read().strip().splitlines()[-1])
print($i, d["ms_per_step"], d.get("dp_overlap_probe_1rank"))
PY
done
