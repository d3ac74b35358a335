"""Tabulate gpurun_out/pmc_hot_*.txt (tools/pmc_hot_kernels.sh): per kernel, per counter, the value PER LAUNCH (sum over the
per-SE / per-XCD records of one dispatch) and derived ratios.  Usage: python tools/pmc_table.py [dir] [launches-per-kernel json]"""
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
tab = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "pmc_hot_*.txt"))):
    for line in open(f):
        m = re.match(r"(.+?)\s+(\w+)\s+calls\s+(\d+)\s+avg\s+([\d.]+)\s+total\s+([\d.]+)", line)
        if not m:
            continue
        name = re.sub(r"\(.*", "", m.group(1)).strip()
        tab[name][m.group(2)] = (int(m.group(3)), float(m.group(5)))
KEYS = ["gemm_nn_a64_kernel<EpiStore, false>", "gemm_nn_a64_kernel<EpiStore, false, true>", "gemm_nn_a64_kernel<EpiSwiGLU, false, true>",
        "gemm_nn_a64_kernel<EpiSwiGLUBwd, false, true>", "gemm_nt_256_kernel<EpiLogpFwd, true, 0, 3, 0, false>",
        "gemm_nt_256_kernel<EpiLogpBwd, true, 0, 3, 0, false>", "gemm_tn_256_kernel<EpiStoreF32, 3, true>", "gemm_tn_256_kernel<EpiStore, 3>",
        "gemm_tn_256_kernel<EpiStore, 3, true>", "attn_fwd2_kernel<128, true>", "attn_fwd2_kernel<128, true, 0>",
        "attn_bwd_dq2_kernel<true>", "attn_bwd_dkv3_kernel<true, 0>", "attn_bwd_dkv2_kernel<true>"]
for k in KEYS:
    if k not in tab:
        continue
    c = tab[k]
    n = c["FETCH_SIZE"][0] if "FETCH_SIZE" in c else 1          # FETCH_SIZE has one record per launch
    print(f"== {k}: {n} launches")
    per = {name: v[1] / n for name, v in c.items()}
    for name in sorted(per):
        print(f"   {name:28s} {per[name]:18.0f}   ({c[name][0] // n} records / launch)")
    g = per.get("GRBM_GUI_ACTIVE", 0) / max(c.get("GRBM_GUI_ACTIVE", (n, 0))[0] // n, 1)       # cycles of one XCD
    if g:
        print(f"   -> kernel cycles (GRBM_GUI_ACTIVE per XCD): {g:.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            print(f"   -> MFMA busy / (1024 SIMDs x cycles): {per['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g):.3f}")
        if "SQ_WAVE_CYCLES" in per:     # quad-cycles
            wc = per["SQ_WAVE_CYCLES"]
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if nm in per:
                    print(f"   -> {nm} / SQ_WAVE_CYCLES: {per[nm] / wc:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in per and "SQ_LDS_IDX_ACTIVE" in per:
            print(f"   -> LDS bank conflict cycles / LDS active cycles: {per['SQ_LDS_BANK_CONFLICT'] / per['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "SQ_LDS_IDX_ACTIVE" in per:
            print(f"   -> LDS active / (256 CUs x cycles): {per['SQ_LDS_IDX_ACTIVE'] / (256 * g):.3f}")
    if "FETCH_SIZE" in per:
        print(f"   -> HBM-side bytes / launch: fetch {2 * per['FETCH_SIZE'] * 1024 / 1e9:.3f} GB (x2 gfx950), write {per.get('WRITE_SIZE', 0) * 1024 / 1e9:.3f} GB")
