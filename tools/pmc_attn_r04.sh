#!/bin/bash
# PMC passes of the attention kernels on the packed-pair rows (bench shape): matrix-pipe busy, wave wait / issue-stall / active
# fractions, LDS activity.  One rocprofv3 run per counter group, --kernel-trace only (never combined with other trace domains).
# Usage (repo root, on the GPU box): RV_ATTN_DKV=5 bash tools/pmc_attn_r04.sh <tag>   -> gpurun_out/pmc_attn_<tag>.txt
set -u
TAG=${1:-r04}
R=$PWD
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for PASS in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
            "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  set -- $PASS; NAME=$1; shift
  rm -rf "/tmp/pa_${TAG}_$NAME"
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d "/tmp/pa_${TAG}_$NAME" -- python "$R/tools/bench_hot_kernels.py" --iters 2 \
      --only attn --attn-case packed > "$R/gpurun_out/pmc_attn_${TAG}_${NAME}.log" 2>&1
  DB=$(find "/tmp/pa_${TAG}_$NAME" -name '*.db' | head -1)
  [ -n "$DB" ] && python "$R/tools/rocpd_pmc.py" "$DB" attn_ > "$R/gpurun_out/pmc_attn_packed_${TAG}${NAME}.txt" 2>&1
done
cd "$R"
python - "$TAG" <<'PY'
import glob, re, collections, sys
tag = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/pmc_attn_packed_{tag}*.txt")):
    for line in open(f):
        m = re.match(r"(.+?)\s+(\w+)\s+calls\s+(\d+)\s+avg\s+([\d.]+)\s+total\s+([\d.]+)", line)
        if m:
            k = re.sub(r"\(.*", "", m.group(1)).strip().split("<")[0]
            tab[k][m.group(2)] = (int(m.group(3)), float(m.group(5)))
out = []
for k, c in sorted(tab.items()):
    n = min(v[0] for v in c.values() if v[0] > 0)
    n = c.get("SQ_WAVES", (n, 0))[0] if "SQ_WAVES" in c else n
    launches = max(1, min(v[0] for v in c.values()) // 8) if False else None
    per = {name: v[1] for name, v in c.items()}
    row = [f"{k:24s}"]
    g = per.get("GRBM_GUI_ACTIVE", 0) / max(c.get("GRBM_GUI_ACTIVE", (1, 0))[0], 1)          # cycles of one XCD record, averaged
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
        recs = c["GRBM_GUI_ACTIVE"][0]                                                          # records = launches x XCDs
        row.append(f"mfma_busy {per['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * per['GRBM_GUI_ACTIVE'] / 8):.3f}")
    wc = per.get("SQ_WAVE_CYCLES")
    if wc:
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if nm in per: row.append(f"{nm[3:]}/wave {per[nm] / wc:.3f}")
    if "SQ_INSTS_VALU" in per and "SQ_INSTS_MFMA" in per: row.append(f"VALU/MFMA insts {per['SQ_INSTS_VALU'] / per['SQ_INSTS_MFMA']:.2f}")
    if "SQ_INSTS_LDS" in per and "SQ_INSTS_MFMA" in per: row.append(f"LDS/MFMA insts {per['SQ_INSTS_LDS'] / per['SQ_INSTS_MFMA']:.2f}")
    if "SQ_LDS_BANK_CONFLICT" in per and "SQ_LDS_IDX_ACTIVE" in per: row.append(f"LDS conflict/active {per['SQ_LDS_BANK_CONFLICT'] / max(per['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
    out.append("  ".join(row))
open(f"gpurun_out/pmc_attn_{tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
