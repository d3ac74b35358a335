"""Generates the straight-line tile bodies of attn_bwd_dkv5_kernel (rlaif-v_amd/csrc/attn_dkv5_*.inc).

What the phase stamps of version 4 showed (profiles/r04_attn_dkv4_phase_profile.log; shader-clock ticks per 64-query tile, wave 0):
64 MFMAs = 2048, yet the tile takes 5700: 16-MFMA phases of 837 / 845 / 957 / 898 instead of 512 (without VALU they ARE 512: a
v_exp_f32 holds the VALU for 16 cycles, so a tile carries ~1500 cycles of VALU and version 4 packs them into two of the four
phases), 325 cycles of exposed LDS latency at the head of every tile, 530 of loop glue, 590 of waits + barrier and - per tile -
1055 of pass prologue / epilogue.  Version 5 keeps version 4's arithmetic layouts and changes the schedule:

  * MFMA order  I: S^T_A / dP^T_A interleaved   II: S^T_B / dP^T_B   III: dV_A dK_A (first 16 queries), dV_A dK_A (last 16)
    IV: the same for B - every stage of sub-tile A is one phase ahead of B, so each phase has VALU work of its own:
        I   the 9 LDS-DMA pieces of tile t + 2, accumulator pre-loads of B            II  exp(S_A), dS_A, packs of A
        III exp(S_A) rest, exp(S_B) first half, packs                                 IV  exp(S_B) rest, dS_B, packs, pre-loads of A(t+1)
  * the row terms are folded into the MFMA accumulator inputs: S^T starts from -lse / scale and dP^T from -delta, so
    P = exp2(c S') is one multiply + one exponential and dS = P dP' one multiply per element (no -log2(e) lse pass, no subtraction);
  * ONE barrier per tile, in front of phase IV: behind it tile t + 1 is complete for everybody, and the first Q / dO row fragments,
    lse and delta of tile t + 1 are read (and its accumulators pre-loaded) under the last MFMAs of tile t - the next tile starts
    with its operands in registers.  The LDS ring has FOUR stages: tile t + 2 is fetched during phase I of tile t, i.e. only after
    the barrier of tile t - 1, which is what guarantees that nobody still reads that buffer (tile t - 2);
  * transposed fragments are requested two groups (8 MFMAs) ahead.

Every LDS read is an explicit asm read and every wait a COUNTED s_waitcnt lgkmcnt(n): this script tracks the issue order of all
reads of the tile and computes n for each use (LDS returns in order; the field is 4 bits: at most 15 younger reads may be in flight
behind a read one waits for - asserted here).  Emits attn_dkv5_body.inc (a tile this wave computes), attn_dkv5_skip.inc (a tile that is
invisible to this wave's 32 keys: DMA, barrier and the pre-loads only) and attn_dkv5_prefetch.inc (loop prologue).
Usage: python tools/gen_attn_dkv5.py
"""
import os

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rlaif-v_amd", "csrc")
X = "AB"


class Sched:
    def __init__(self):
        self.lines = []
        self.issued = 0
        self.last = {}
        self.done = 0
        self.max_wait = 0

    def emit(self, s):
        self.lines.append("          " + s)

    def read(self, name, stmt, n=1):
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

    def stmt(self, s):
        self.emit(s)
        self.emit("__builtin_amdgcn_sched_barrier(0);")


def rq(s, x, ks, nxt=False):
    s.read(f"Q{X[x]}{ks}{'n' if nxt else ''}", f"qfrag_load<{x * 8 + ks}>(rb[{ks}], {x * 8192});")


def rf(s, x, ks, nxt=False):
    s.read(f"F{X[x]}{ks}{'n' if nxt else ''}", f"fo{X[x]}[{ks}] = ds_read_b128_asm(rb[{ks}], {x * 8192 + 16384});")


def rl(s, x, j, nxt=False):      # -lse / scale of queries 8 j + 4 half .. + 3 of sub-tile x: the S^T accumulator input, rows 4 j .. 4 j + 3
    s.read(f"L{X[x]}{j}{'n' if nxt else ''}", f"acc_quarter(s{X[x]}, {j}, ds_read_f32x4_asm(lh, {x * 128 + j * 32}));")


def rd(s, x, j, nxt=False):      # -delta: the dP^T accumulator input
    s.read(f"D{X[x]}{j}{'n' if nxt else ''}", f"acc_quarter(p{X[x]}, {j}, ds_read_f32x4_asm(lh, {256 + x * 128 + j * 32}));")


# transposed-fragment groups in MFMA order: (sub-tile, 16-query half, is_dO); group g feeds MFMAs 32 + 4 g .. 32 + 4 g + 3
GROUPS = [(x, h, isdo) for x in (0, 1) for h in (0, 1) for isdo in (1, 0)]


def rt(s, g, e):
    x, h, isdo = GROUPS[g]
    off = x * 32 * 256 + isdo * 16384 + h * 16 * 256
    s.read(f"T{g}_{e}", f"tr[{g % 3}][{e}] = __builtin_shufflevector(ds_tr16_b64_asm(tb[0][{e}], {off}), "
                        f"ds_tr16_b64_asm(tb[1][{e}], {off}), 0, 1, 2, 3, 4, 5, 6, 7);", n=2)


def prefetch(s):
    """lse and delta, then row fragments 0..3 of Q (-> AGPR) and dO, of sub-tile A of the NEXT tile (rb / lh already point into its
    buffer).  The dQ kernel publishes -lse / scale and -delta, so those reads land straight in the accumulator registers of
    S^T_A and dP^T_A.  Returns 8 read steps (two reads each)."""
    steps = []
    for j in range(4):
        steps.append(lambda j=j: (rl(s, 0, j, True), rd(s, 0, j, True)))
    for ks in range(4):
        steps.append(lambda ks=ks: (rq(s, 0, ks, True), rf(s, 0, ks, True)))
    return steps


def body():
    s = Sched()
    s.emit("// ---- GENERATED by tools/gen_attn_dkv5.py: do not edit; the waits are counted against THIS issue order ----")
    # registers on entry (previous tile's phase IV or the loop prologue): Q_A[0..3] in AGPRs, foA[0..3], sA = -lse / scale, pA = -delta
    s.emit("PROF(0);")
    for j in range(4):                       # lse / delta of sub-tile B first: its pre-loads run in the middle of phase I
        rl(s, 1, j)
        rd(s, 1, j)
    pf = None
    for j in range(64):
        # ------------------------------------------------------------------ in front of the MFMA
        if j == 16:
            s.emit("PROF(1);")
        if j == 32:
            s.emit("PROF(2);")
        if j == 48:
            s.emit("PROF(3);")
            s.emit("// ---- the ONE barrier of the tile: tile t + 1 landed (this wave's pieces, then everybody's); row-read bases -> next buffer")
            s.emit("wait_all_but_newest();")
            s.emit("PROF(5);")
            s.emit("__builtin_amdgcn_s_barrier();")
            s.emit("PROF(6);")
            s.stmt("flip_rows();")
            pf = prefetch(s)
        if j < 32:
            x, k, isp = (0 if j < 16 else 1), (j % 16) >> 1, j & 1
            if j == 8:
                s.need("FA7")                # the rest of sub-tile A's row fragments, as one batch (requested in slots 0..3)
            if j == 16:
                s.need("FB7")                # sub-tile B's row fragments, as one batch (requested in slots 4..11)
            if isp:
                s.emit(f"smfma<{8 + k}>(p{X[x]}, fo{X[x]}[{k}]);")
            else:
                s.emit(f"smfma_q<{x * 8 + k}, {k}, false>(s{X[x]});")
        else:
            g, e = (j - 32) >> 2, j & 3
            x, h, isdo = GROUPS[g]
            if e == 0:
                s.need(f"T{g}_3")            # a transposed group lands as a whole, two groups (8 MFMAs) after it was requested
            fn = "acc_mfma" if j in (48, 56) else "acc_mfma_nn"      # 48 / 56: the pack was written in the gap right before
            if isdo:
                s.emit(f"{fn}<{4 + e}>(tr[{g % 3}][{e}], BF(pf{X[x]}{h}));")
            else:
                s.emit(f"{fn}<{e}>(tr[{g % 3}][{e}], BF(ds{X[x]}{h}));")
        s.emit("__builtin_amdgcn_sched_barrier(0);")
        # ------------------------------------------------------------------ LDS reads behind it
        if j < 4:                            # rest of sub-tile A's row fragments (needed from slot 8 on)
            rq(s, 0, 4 + j)
            rf(s, 0, 4 + j)
        if 4 <= j < 12:                      # sub-tile B's row fragments (needed from slot 16 on)
            rq(s, 1, j - 4)
            rf(s, 1, j - 4)
        if 20 <= j < 28:                     # transposed groups 0 and 1
            rt(s, (j - 20) >> 2, j & 3)
        if 32 <= j < 56:                     # group g + 2 while group g multiplies
            rt(s, ((j - 32) >> 2) + 2, j & 3)
        # ------------------------------------------------------------------ VALU / DMA behind it
        if 1 <= j <= 9:
            s.stmt(f"issue_piece(hq, tn, bufn, {j - 1});")
        if j == 3:
            s.stmt("mask_flags();")
        if j == 17:
            s.stmt("if (need_mask_a) kmask.apply(sA, qs0 + 4 * half);")
        if 17 <= j <= 24:
            s.stmt(f"exps(sA, {j - 17}, {j - 16});")
        if j == 25:
            s.stmt("pack4(pfA0, sA, 0);")
        if j == 26:
            s.stmt("dsmul(pA, sA, 0, 8);")
        if j == 27:
            s.stmt("pack4(dsA0, pA, 0);")
        if 28 <= j <= 35:
            s.stmt(f"exps(sA, {j - 20}, {j - 19});")
        if j == 36:
            s.stmt("pack4(pfA1, sA, 8);")
        if j == 37:
            s.stmt("dsmul(pA, sA, 8, 16);")
        if j == 38:
            s.stmt("pack4(dsA1, pA, 8);")
            s.stmt("if (need_mask_b) kmask.apply(sB, qs0 + 32 + 4 * half);")
        if 39 <= j <= 46:
            s.stmt(f"exps(sB, {j - 39}, {j - 38});")
        if j == 47:
            s.stmt("pack4(pfB0, sB, 0);")
        if j == 48:
            s.stmt("dsmul(pB, sB, 0, 8);")
        if j == 49:
            s.stmt("pack4(dsB0, pB, 0);")
        if 48 <= j <= 55:
            s.stmt(f"exps(sB, {j - 40}, {j - 39});")
        if j == 55:
            s.stmt("pack4(pfB1, sB, 8);")
        if j == 57:
            s.stmt("dsmul(pB, sB, 8, 16);")
        if j == 58:
            s.stmt("pack4(dsB1, pB, 8);")
        if 54 <= j <= 61:                    # next tile's lse / delta, then its first row fragments: two reads per slot
            pf[j - 54]()
    s.need("FA3n")                           # the next tile starts with its first operands in registers
    s.emit("PROF(4);")
    s.stmt("flip_tr();")
    s.emit(f"// ---- end of the generated tile body: {s.issued} LDS reads, deepest counted wait lgkmcnt({s.max_wait}) ----")
    return s


def skip():
    s = Sched()
    s.emit("// ---- GENERATED by tools/gen_attn_dkv5.py: a tile this wave's 32 keys do not see - its share of the DMA, the barrier,")
    s.emit("// ---- and the pre-loads of the following tile (which it may compute)")
    for j in range(9):
        s.stmt(f"issue_piece(hq, tn, bufn, {j});")
    s.emit("wait_all_but_newest();")
    s.emit("__builtin_amdgcn_s_barrier();")
    s.stmt("flip_rows();")
    for fn in prefetch(s):
        fn()
    s.need("FA3n")
    s.stmt("flip_tr();")
    return s


def pre():
    s = Sched()
    s.emit("// ---- GENERATED by tools/gen_attn_dkv5.py: loop prologue - pre-loads of the first tile (bases point into its buffer)")
    for fn in prefetch(s):
        fn()
    s.need("FA3n")
    return s


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=CSRC, help="directory the three attn_dkv5_*.inc files are written to (default: csrc/)")
    outdir = ap.parse_args().out
    for name, s in (("body", body()), ("skip", skip()), ("prefetch", pre())):
        out = os.path.join(outdir, f"attn_dkv5_{name}.inc")
        open(out, "w").write("\n".join(s.lines) + "\n")
        print(out, s.issued, "reads, max wait", s.max_wait)


if __name__ == "__main__":
    main()
