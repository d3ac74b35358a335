#!/bin/bash
# Socket power and shader clock sampled while the full DPO step runs (evidence for "the step is energy-bound").
# Usage (GPU box, repo root): bash tools/exp_step_power.sh  -> gpurun_out/step_power.log
mkdir -p gpurun_out
LOG=gpurun_out/step_power.log
: > $LOG
python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-dp-probe > gpurun_out/step_power_bench.log 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  T=$(date +%s.%N | cut -c1-14)
  S=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | sed -E 's/.*sclk clock level: [0-9]+: \(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' ')
  echo "$T $S" >> $LOG
  sleep 0.4
done
tail -1 gpurun_out/step_power_bench.log | cut -c1-200
python3 - <<'PY'
import re
rows=[]
for l in open("gpurun_out/step_power.log"):
    m=re.search(r"sclk (\d+).*W ([\d.]+)", l)
    if m: rows.append((int(m.group(1)), float(m.group(2))))
busy=[r for r in rows if r[1] > 900]
print(f"{len(rows)} samples, {len(busy)} above 900 W")
if busy:
    print("while training: mean power %.0f W, mean sclk %.0f MHz (min %d, max %d)" % (sum(r[1] for r in busy)/len(busy), sum(r[0] for r in busy)/len(busy), min(r[0] for r in busy), max(r[0] for r in busy)))
PY
