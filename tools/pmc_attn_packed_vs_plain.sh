#!/bin/bash
# PMC passes of the three attention kernels on the packed-pair rows and on plain causal rows SEPARATELY (same 64-key tile
# count, the packed forward is ~19 % slower: why?).  One rocprofv3 run per (case, counter group), --kernel-trace only.
# Writes gpurun_out/pmc_attn_<case>_<pass>.txt and a merged table gpurun_out/pmc_attn_packed_vs_plain.txt.
set -u
R=$PWD
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for CASE in packed plain; do
  for PASS in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
              "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
              "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
              "fetch FETCH_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $PASS; NAME=$1; shift
    rm -rf "/tmp/pa_${CASE}_$NAME"
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d "/tmp/pa_${CASE}_$NAME" -- python "$R/tools/bench_hot_kernels.py" --iters 2 \
        --only attn --attn-case $CASE > "$R/gpurun_out/pmc_attn_${CASE}_${NAME}.log" 2>&1
    DB=$(find "/tmp/pa_${CASE}_$NAME" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$R/tools/rocpd_pmc.py" "$DB" attn_ > "$R/gpurun_out/pmc_attn_${CASE}_${NAME}.txt" 2>&1
  done
done
cd "$R"
python - <<'PY'
import glob, re, collections
tab = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc_attn_*_*.txt")):
    case = f.split("pmc_attn_")[1].split("_")[0]
    for line in open(f):
        m = re.match(r"(.+?)\s+(\w+)\s+calls\s+(\d+)\s+avg\s+([\d.]+)\s+total\s+([\d.]+)", line)
        if m:
            k = re.sub(r"\(.*", "", m.group(1)).strip().split("<")[0]
            tab[(k, case)][m.group(2)] = (int(m.group(3)), float(m.group(5)))
out = []
for (k, case), c in sorted(tab.items()):
    n = c.get("FETCH_SIZE", (1, 0))[0]
    per = {name: v[1] / n for name, v in c.items()}
    g = per.get("GRBM_GUI_ACTIVE", 0) / max(c.get("GRBM_GUI_ACTIVE", (n, 0))[0] // n, 1)
    row = [f"{k:22s} {case:6s} launches {n}"]
    if g:
        row.append(f"cycles/XCD {g:9.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per: row.append(f"mfma_busy {per['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g):.3f}")
    wc = per.get("SQ_WAVE_CYCLES")
    if wc:
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if nm in per: row.append(f"{nm[3:]}/wave {per[nm] / wc:.3f}")
        row.append(f"wave_cycles {wc:.3e}")
    for nm in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if nm in per: row.append(f"{nm[3:]} {per[nm]:.3e}")
    if "FETCH_SIZE" in per: row.append(f"fetch_GB {2 * per['FETCH_SIZE'] * 1024 / 1e9:.3f}")
    if "TCC_HIT_sum" in per: row.append(f"L2_hit {per['TCC_HIT_sum'] / (per['TCC_HIT_sum'] + per['TCC_MISS_sum']):.3f}")
    out.append("  ".join(row))
open("gpurun_out/pmc_attn_packed_vs_plain.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
