"""Micro-benchmark of the step's dominant kernels at the bench shape (8 packed pairs, 27,664 tokens): the NN / TN GEMMs of
one decoder layer and the packed-pair attention forward / backward.  Also the target of the PMC passes
(tools/pmc_hot_kernels.sh).  Usage: python tools/bench_hot_kernels.py [--iters 5] [--only gemm|attn]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def packed_attention_inputs(dev, B=8, H=32, hd=128, shared=638, tail=1410):
    """Rows shaped like the bench's packed pairs: [shared | chosen tail | rejected tail]."""
    L = shared + 2 * tail
    d = H * hd
    qkv = (torch.randn(B * L, 3 * d, device=dev) * 0.5).to(BF)
    do = (torch.randn(B * L, d, device=dev) * 0.5).to(BF)
    seg_sh = torch.full((B,), shared, dtype=torch.int32, device=dev)
    seg_e1 = torch.full((B,), shared + tail, dtype=torch.int32, device=dev)
    return qkv, do, (seg_sh, seg_e1), L, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--attn-case", default="", help="packed | plain: run only that attention case (PMC passes per case)")
    ap.add_argument("--attn-rounds", type=int, default=1,
                    help="repeat the attention measurements this many times and print every round: the FIRST kernel measured after "
                         "process start runs at a lower shader clock (the clock ramps up over the first ~100 ms of load) - round 3 "
                         "found that this, not the layout, made the packed forward look 19 %% slower than the plain one")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R = 27664
    if a.only in ("", "gemm"):
        for name, N, K in [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]:
            x = torch.randn(R, K, device=dev).to(BF)
            wT = (torch.randn(K, N, device=dev) * 0.02).to(BF)
            out = torch.empty(R, N, device=dev, dtype=BF)
            ms = timeit(lambda: ops.gemm_nn(x, wT, out=out), a.iters)
            print(f"nn {name:8s} {R}x{N}x{K}: {ms:.3f} ms {2.0 * R * N * K / ms / 1e9:7.1f} TF/s", flush=True)
        # fused-epilogue instantiations of the same NN main loop next to the plain kernel on the same shape
        x = torch.randn(R, 4096, device=dev).to(BF)
        wguT = (torch.randn(4096, 22016, device=dev) * 0.02).to(BF)
        ms = timeit(lambda: ops.linear_swiglu(x, wguT), a.iters)
        print(f"nn_swiglu     {R}x22016x4096: {ms:.3f} ms {2.0 * R * 22016 * 4096 / ms / 1e9:7.1f} TF/s", flush=True)
        dy = torch.randn(R, 4096, device=dev).to(BF)
        wdown = (torch.randn(4096, 11008, device=dev) * 0.02).to(BF)
        gu = torch.randn(R, 22016, device=dev).to(BF)
        ms = timeit(lambda: ops.linear_swiglu_bwd(dy, wdown, gu), a.iters)
        print(f"nn_swiglu_bwd {R}x11008x4096: {ms:.3f} ms {2.0 * R * 11008 * 4096 / ms / 1e9:7.1f} TF/s", flush=True)
        out = torch.empty(R, 11008, device=dev, dtype=BF)
        ms = timeit(lambda: ops.gemm_nn(dy, wdown, out=out), a.iters)
        print(f"nn plain      {R}x11008x4096: {ms:.3f} ms {2.0 * R * 11008 * 4096 / ms / 1e9:7.1f} TF/s", flush=True)
        del x, wguT, dy, wdown, gu, out
        n_sel, V = 22544, 32000                     # 8 pairs x 2 x 1409 target rows
        h = torch.randn(n_sel, 4096, device=dev).to(BF)
        wl = (torch.randn(V, 4096, device=dev) * 0.02).to(BF)
        tgt = torch.randint(0, V, (n_sel,), device=dev, dtype=torch.int32)
        logp, lse = ops.lmhead_logp_fwd(h, wl, tgt, n_sel)
        ms = timeit(lambda: ops.lmhead_logp_fwd(h, wl, tgt, n_sel), a.iters)
        print(f"lmhead_fwd {n_sel}x{V}x4096: {ms:.3f} ms {2.0 * n_sel * V * 4096 / ms / 1e9:7.1f} TF/s", flush=True)
        coef = torch.full((n_sel,), -0.01, device=dev)
        dl = torch.zeros(n_sel, V, dtype=BF, device=dev)
        ms = timeit(lambda: ops.lmhead_logp_bwd(h, wl, tgt, lse, coef, n_sel, out=dl), a.iters)
        print(f"lmhead_bwd {n_sel}x{V}x4096: {ms:.3f} ms {2.0 * n_sel * V * 4096 / ms / 1e9:7.1f} TF/s", flush=True)
        del h, wl, dl
        for name, I, J in [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("wgu", 22016, 4096), ("wdown", 4096, 11008)]:
            p = torch.randn(R, I, device=dev).to(BF)
            q = torch.randn(R, J, device=dev).to(BF)
            out = torch.empty(I, J, device=dev, dtype=BF)
            ms = timeit(lambda: ops.gemm_tn(p, q, out=out), a.iters)
            print(f"tn {name:8s} {R}: {I}x{J}: {ms:.3f} ms {2.0 * R * I * J / ms / 1e9:7.1f} TF/s", flush=True)
    if a.only in ("", "attn"):
        B, H, hd = 8, 32, 128
        qkv, do, seg, L, d = packed_attention_inputs(dev, B, H, hd)
        # algorithmic (q, k) pairs of one packed row: shared causal + each tail sees shared and itself causally
        sh, tl = int(seg[0][0]), int(seg[1][0] - seg[0][0])
        pairs = sh * (sh + 1) / 2 + 2 * (tl * sh + tl * (tl + 1) / 2)
        fl_f = 4.0 * B * H * pairs * hd
        if a.attn_case == "plain":
            del qkv, do
            qkv = do = None
    for _round in range(a.attn_rounds):
      _attn_round(a, dev, locals())


def _attn_round(a, dev, env):
    BF = torch.bfloat16
    if a.only in ("", "attn") and a.attn_case != "plain":
        qkv, do, seg, L, d, B, H, hd, sh, tl, fl_f = (env[k] for k in ("qkv", "do", "seg", "L", "d", "B", "H", "hd", "sh", "tl", "fl_f"))
        o, lse = ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, seg=seg)
        out = torch.empty_like(o)
        t_f = timeit(lambda: ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=out, seg=seg), a.iters)
        dqkv = torch.empty_like(qkv)
        t_b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv, seg=seg), a.iters)
        print(f"attn packed L={L} (shared {sh}, tails {tl}): fwd {t_f:.3f} ms ({fl_f / t_f / 1e9:.0f} TF/s algorithmic)  "
              f"bwd (delta+dq+dkv) {t_b:.3f} ms ({2.5 * fl_f / t_b / 1e9:.0f} TF/s algorithmic)", flush=True)
    if a.only in ("", "attn") and a.attn_case != "packed":
        B, H, hd, d = 8, 32, 128, 4096
        # plain causal rows of the reference layout for comparison (16 x 2048)
        S2, L2 = 16, 2048
        qkv2 = (torch.randn(S2 * L2, 3 * d, device=dev) * 0.5).to(BF)
        do2 = (torch.randn(S2 * L2, d, device=dev) * 0.5).to(BF)
        o2, lse2 = ops.attn_fwd(qkv2, S2, L2, H, hd, True, 0, d, 2 * d)
        t_f2 = timeit(lambda: ops.attn_fwd(qkv2, S2, L2, H, hd, True, 0, d, 2 * d, out=o2), a.iters)
        dq2 = torch.empty_like(qkv2)
        t_b2 = timeit(lambda: ops.attn_bwd(qkv2, o2, do2, lse2, S2, L2, H, hd, True, 0, d, 2 * d, dqkv=dq2), a.iters)
        fl2 = 4.0 * S2 * H * (L2 * (L2 + 1) / 2) * hd
        print(f"attn plain  16 x 2048: fwd {t_f2:.3f} ms ({fl2 / t_f2 / 1e9:.0f} TF/s)  bwd {t_b2:.3f} ms ({2.5 * fl2 / t_b2 / 1e9:.0f} TF/s)",
              flush=True)


if __name__ == "__main__":
    main()
