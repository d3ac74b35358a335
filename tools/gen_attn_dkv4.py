"""Generates rlaif-v_amd/csrc/attn_dkv4_body.inc: the straight-line body of ONE 64-query tile of attn_bwd_dkv4_kernel.

Version 4 of the dK/dV kernel keeps version 3's decomposition, LDS image, operand layouts and arithmetic and changes WHO WAITS
FOR WHOM inside a tile.  Version 3 runs the two 32-query sub-tiles of a tile one after the other, each as a dependent chain
S^T -> exp -> dP^T -> dS -> dV / dK inside the single wave of a SIMD: PMC says matrix pipe busy 0.31, issue stalls 0.37
(profiles/r03_pmc_attn_packed_vs_plain_final.txt).  Here the two sub-tiles A and B are IN FLIGHT TOGETHER and the tile is four
phases of 16 MFMAs whose VALU / LDS work always belongs to ANOTHER stage than the MFMAs it sits between:

    P1  S^T_A, S^T_B interleaved (two independent chains: no half-rate dependent issue, no split partial sums)
          | reads: rest of the Q row fragments, first dO row fragments;  VALU: -log2(e) x lse for both sub-tiles
    P2  dP^T_A, dP^T_B interleaved
          | reads: rest of dO, delta_A, first dO^T fragments;  VALU: masks, exp(S_A), exp(S_B), pack P_A
    P3  dV_A (8), dK_A (8)
          | reads: transposed fragments one group ahead, delta_B;  VALU: pack P_B, dS_A, pack dS_A, dS_B, pack dS_B
    P4  dV_B (8), dK_B (8)
          | reads: transposed fragments one group ahead;  the 9 LDS-DMA pieces of tile t + 2 (three-stage ring)

Every LDS read is an explicit asm read and every wait a COUNTED s_waitcnt lgkmcnt(n): this script tracks the issue order of
all reads of the tile and computes n for each use (LDS returns in order; the field is 4 bits, so at most 15 younger reads may be
in flight behind a read one waits for - asserted here).  Usage: python tools/gen_attn_dkv4.py
"""
import os

import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rlaif-v_amd", "csrc")
OUT = os.path.join(CSRC, "attn_dkv4_body.inc")
# Ablation bodies (experiment builds -DRV_DKV4_ABL=n, results WRONG by construction; timing deltas price the stages):
#   1 no transposing reads, 2 no row / lse / delta reads, 3 no VALU, 4 no MFMAs, 5 no LDS reads at all, 6 MFMAs only (+ DMA),
#   7 no LDS-DMA of the following tiles, 8 nothing but DMA + barrier
ABL = {0: set(), 1: {"tr"}, 2: {"row"}, 3: {"valu"}, 4: {"mfma"}, 5: {"tr", "row"}, 6: {"tr", "row", "valu"}, 7: {"dma"},
       8: {"tr", "row", "valu", "mfma"}}


class Sched:
    def __init__(self, skip=()):
        self.skip = set(skip)
        self.lines = []
        self.issued = 0            # LDS reads issued so far
        self.last = {}             # read name -> index (1-based) of its last read
        self.done = 0              # reads known complete (index)
        self.max_wait = 0

    def emit(self, s):
        self.lines.append("          " + s)

    def read(self, name, stmt, n=1, kind="row"):
        if kind in self.skip:
            self.last[name] = self.issued        # never issued: any wait for it is already satisfied
            return
        self.emit(stmt)
        self.issued += n
        self.last[name] = self.issued

    def need(self, *names):
        idx = max(self.last[n] for n in names)
        if idx <= self.done:
            return
        cnt = self.issued - idx
        assert 0 <= cnt <= 15, (names, cnt)
        self.max_wait = max(self.max_wait, cnt)
        self.emit(f'asm volatile("s_waitcnt lgkmcnt({cnt})" ::: "memory");   // {", ".join(names)} landed')
        self.emit("__builtin_amdgcn_sched_barrier(0);")
        self.done = idx

    def mfma(self, stmt):
        if "mfma" not in self.skip:
            self.emit(stmt)

    def valu(self, stmt, kind="valu"):
        if kind in self.skip:
            return
        self.emit(stmt)
        self.emit("__builtin_amdgcn_sched_barrier(0);")

    def fence(self):
        self.emit("__builtin_amdgcn_sched_barrier(0);")


def generate(abl=0):
    s = Sched(ABL[abl])
    X = "AB"

    def rq(x, ks):      # Q row fragment ks of sub-tile x -> AGPR slot x*8 + ks
        s.read(f"Q{X[x]}{ks}", f"qfrag_load<{x * 8 + ks}>(rb[{ks}], {x * 8192});")

    def rf(x, ks):      # dO row fragment
        s.read(f"F{X[x]}{ks}", f"fo{X[x]}[{ks}] = ds_read_b128_asm(rb[{ks}], {x * 8192 + 16384});")

    def rl(x, j):       # lse of queries 8 j + 4 half .. + 3 of sub-tile x
        s.read(f"L{X[x]}{j}", f"ls{X[x]}[{j}] = ds_read_f32x4_asm(lh, {x * 128 + j * 32});")

    def rd(x, j):
        s.read(f"D{X[x]}{j}", f"de{X[x]}[{j}] = ds_read_f32x4_asm(lh, {256 + x * 128 + j * 32});")

    # transposed-fragment groups of the dV / dK stream: (sub-tile, is_dO, 16-query half); group g is consumed by MFMAs 4g .. 4g+3
    groups = [(x, isdo, h) for x in (0, 1) for isdo in (1, 0) for h in (0, 1)]

    def rt(g, e):
        x, isdo, h = groups[g]
        off = x * 32 * 256 + isdo * 16384 + h * 16 * 256
        s.read(f"T{g}_{e}", f"tr[{g & 1}][{e}] = __builtin_shufflevector(ds_tr16_b64_asm(tb[0][{e}], {off}), "
                            f"ds_tr16_b64_asm(tb[1][{e}], {off}), 0, 1, 2, 3, 4, 5, 6, 7);", n=2, kind="tr")

    s.emit("// ---- GENERATED by tools/gen_attn_dkv4.py: do not edit; the waits are counted against THIS issue order ----")
    s.emit("PROF(0);")
    # prologue: first half of the Q row fragments of both sub-tiles, lse of both
    for ks in range(4):
        rq(0, ks)
        rq(1, ks)
    for j in range(4):
        rl(0, j)
    for j in range(4):
        rl(1, j)
    # ---- P1: S^T_A / S^T_B
    s.emit("// ---- P1: S^T of both sub-tiles, interleaved")
    for j in range(16):
        x, ks = j & 1, j >> 1
        s.need(f"Q{X[x]}{ks}")
        s.mfma(f"smfma_q<{x * 8 + ks}, {ks}, {'true' if ks == 0 else 'false'}>(s{X[x]});")
        s.fence()
        if j < 8:
            rq(j & 1, 4 + (j >> 1))
        elif j < 16:
            rf((j - 8) & 1, (j - 8) >> 1)
        if j in (4, 5):
            s.need("LA3")
            s.valu(f"prescale(lsA, {2 * (j - 4)}, {2 * (j - 4) + 2});")
        if j in (6, 7):
            s.need("LB3")
            s.valu(f"prescale(lsB, {2 * (j - 6)}, {2 * (j - 6) + 2});")
    s.emit("PROF(1);")
    # ---- P2: dP^T_A / dP^T_B with the exponentials
    s.emit("// ---- P2: dP^T of both sub-tiles, interleaved; exponentials of S^T underneath")
    # (the last S^T_A MFMA is P1 slot 14: VALU may read sA from three MFMAs later on = after P2 slot 1)
    expA = {j: (2 * (j - 2), 2 * (j - 1)) for j in range(2, 10)}            # slots 2..9, two elements each
    expB = {j: (2 * (j - 10), 2 * (j - 9)) for j in range(10, 16)}          # slots 10..15: elements 0..11; 12..15 follow in P3
    for j in range(16):
        x, ks = j & 1, j >> 1
        s.need(f"F{X[x]}{ks}")
        s.mfma(f"smfma_v<{8 + ks}, {'true' if ks == 0 else 'false'}>(p{X[x]}, fo{X[x]}[{ks}]);")
        s.fence()
        if j < 8:
            rf(j & 1, 4 + (j >> 1))
        elif j < 12:
            rd(0, j - 8)
        else:
            rt(0, j - 12)
        if j == 1:
            s.valu("if (need_mask_a) kmask.apply(sA, qs0 + 4 * half);")
        if j == 9:
            s.valu("if (need_mask_b) kmask.apply(sB, qs0 + 32 + 4 * half);")
        if j in expA:
            s.valu(f"exps(sA, lsA, {expA[j][0]}, {expA[j][1]});")
        if j in expB:
            s.valu(f"exps(sB, lsB, {expB[j][0]}, {expB[j][1]});")
        if 10 <= j <= 13:
            q = j - 10         # two of the eight cvt_pk of P_A per slot: fragment q >> 1, dwords 2 (q & 1), 2 (q & 1) + 1
            s.valu(f"pack2(pfA{q >> 1}, sA, {(q >> 1) * 8}, {(q & 1) * 2});")
    s.emit("PROF(2);")
    # ---- P3 / P4: dV, dK
    s.emit("// ---- P3: dV_A, dK_A; dS of both sub-tiles underneath.  P4: dV_B, dK_B; the LDS-DMA of tile t + 2 underneath")
    dsA = {2: (0, 3), 3: (3, 6), 4: (6, 9), 5: (9, 12), 6: (12, 15), 7: (15, 16)}
    dsB = {10: (0, 3), 11: (3, 6), 12: (6, 9), 13: (9, 12), 14: (12, 15), 15: (15, 16)}
    for j in range(32):
        g, e = j >> 2, j & 3
        x, isdo, h = groups[g]
        s.need(f"T{g}_{e}")
        if isdo:
            s.mfma(f"acc_mfma<{4 + e}>(tr[{g & 1}][{e}], BF(pf{X[x]}{h}));")
        else:
            s.mfma(f"acc_mfma<{e}>(tr[{g & 1}][{e}], BF(ds{X[x]}{h}));")
        s.fence()
        if g + 1 < 8:
            rt(g + 1, e)
        if j < 4:
            rd(1, j)
        if j == 0:
            s.valu("exps(sB, lsB, 12, 14);")
            s.valu("pack4(pfB0, sB, 0);")
        if j == 1:
            s.valu("exps(sB, lsB, 14, 16);")
        if j in dsA:
            if j == 2:
                s.need("DA3")
            s.valu(f"dsmul(pA, sA, deA, {dsA[j][0]}, {dsA[j][1]});")
        if j == 7:
            s.valu("pack4(dsA0, pA, 0);")
        if j == 8:
            s.valu("pack4(pfB1, sB, 8);")
        if j == 9:
            s.valu("pack4(dsA1, pA, 8);")
        if j in dsB:
            if j == 10:
                s.need("DB3")
            s.valu(f"dsmul(pB, sB, deB, {dsB[j][0]}, {dsB[j][1]});")
        if j == 15:
            s.valu("pack4(dsB0, pB, 0);")
        if j == 16:
            s.valu("pack4(dsB1, pB, 8);")
        if 19 <= j <= 27:
            s.valu(f"issue_piece(hq, tn, bufn, {j - 19});", kind="dma")
        if j == 15:
            s.emit("PROF(3);")
        if j == 31:
            s.emit("PROF(4);")
    s.emit(f"// ---- end of the generated tile body: {s.issued} LDS reads, deepest counted wait lgkmcnt({s.max_wait}) ----")
    out = OUT if abl == 0 else OUT.replace(".inc", f"_abl{abl}.inc")
    open(out, "w").write("\n".join(s.lines) + "\n")
    print(out, s.issued, "reads, max wait", s.max_wait)


if __name__ == "__main__":
    generate(0)
    if "--ablations" in sys.argv:
        for a in sorted(ABL):
            if a:
                generate(a)
