"""The rank-64 side of a LoRA decoder layer at production widths (config 5: L = 4096, 4 pairs): every skinny launch of one
layer's forward + backward against the HBM floor of the operand it streams.  Usage: python tools/exp_lora_skinny.py [--rows 29000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16
HBM = 5.0e12          # a streaming kernel's practical rate on this part (bytes / s), for the "floor" column


def timeit(fn, iters=10, warmup=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=29000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, r = a.rows, 64
    g = torch.Generator(device=dev).manual_seed(0)
    tot = {}
    print(f"rows {M}; floor = streamed operand bytes / {HBM / 1e12:.1f} TB/s")
    for name, in_w, outs in (("qkv", 4096, (4096, 4096, 4096)), ("o", 4096, (4096,)), ("gate|up", 4096, (11008, 11008)),
                             ("down", 11008, (4096,))):
        G, out_w = len(outs), sum(outs)
        x = torch.randn(M, in_w, device=dev, generator=g).to(BF)
        dy = torch.randn(M, out_w, device=dev, generator=g).to(BF)
        A = (torch.randn(G * r, in_w, device=dev, generator=g) * 0.02).to(BF)
        BT = (torch.randn(r, out_w, device=dev, generator=g) * 0.02).to(BF)
        t = torch.randn(M, G * r, device=dev, generator=g).to(BF)
        dt = torch.empty(M, G * r, device=dev, dtype=BF)
        gA, gB = torch.empty_like(A), torch.empty(out_w, r, device=dev, dtype=BF)
        xd = ops.dropout(x, 0.05, 3)
        cols = [0]
        for o in outs:
            cols.append(cols[-1] + o)

        def f_dt():
            for gi in range(G):
                ops.gemm_nt(dy[:, cols[gi]:cols[gi + 1]], BT[:, cols[gi]:cols[gi + 1]], out=dt[:, gi * r:(gi + 1) * r], alpha=0.25)

        def f_dB():
            for gi in range(G):
                ops.gemm_tn_skinny(dy[:, cols[gi]:cols[gi + 1]], t[:, gi * r:(gi + 1) * r], out=gB[cols[gi]:cols[gi + 1]])
        side = [torch.cuda.Stream() for _ in range(G)]

        def f_dt_par():          # the G group launches side by side (what ONE batched launch would look like to the memory system)
            main = torch.cuda.current_stream()
            for gi in range(G):
                side[gi].wait_stream(main)
                with torch.cuda.stream(side[gi]):
                    ops.gemm_nt(dy[:, cols[gi]:cols[gi + 1]], BT[:, cols[gi]:cols[gi + 1]], out=dt[:, gi * r:(gi + 1) * r], alpha=0.25)
            for gi in range(G):
                main.wait_stream(side[gi])
        if G > 1:
            ms = timeit(f_dt_par)
            print(f"{name:8s} dt, {G} groups on {G} streams: {ms:7.3f} ms", flush=True)
        rows = [("dropout(x)", lambda: ops.dropout(x, 0.05, 3), 2 * M * in_w * 2),
                ("t = xd A^T", lambda: ops.gemm_nt(xd, A, alpha=0.25), M * in_w * 2),
                ("dt = dy B", f_dt, M * out_w * 2),
                ("dA = dt^T xd", lambda: ops.gemm_tn_skinny(dt, xd, out=gA), M * in_w * 2),
                ("dB = dy^T t", f_dB, M * out_w * 2)]
        for label, fn, nbytes in rows:
            ms = timeit(fn)
            floor = nbytes / HBM * 1e3
            tot.setdefault(label, [0.0, 0.0])
            tot[label][0] += ms
            tot[label][1] += floor
            print(f"{name:8s} {label:14s} {ms:7.3f} ms   floor {floor:6.3f} ms   x{ms / floor:4.1f}", flush=True)
        del x, dy, xd, t, dt
    print("per layer:")
    s0 = s1 = 0.0
    for label, (ms, fl) in tot.items():
        print(f"  {label:14s} {ms:7.3f} ms   floor {fl:6.3f} ms   x{ms / fl:4.1f}")
        s0, s1 = s0 + ms, s1 + fl
    print(f"  sum            {s0:7.3f} ms   floor {s1:6.3f} ms  -> {32 * s0:.1f} ms per 32-layer step (floor {32 * s1:.1f})")


if __name__ == "__main__":
    main()
