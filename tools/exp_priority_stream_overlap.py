#!/usr/bin/env python
"""Round-6 experiment: can the weight-gradient (TN) GEMMs of backward fill the idle CUs in the LAST ROUND of the input-gradient
(NN) GEMMs' tiles (N = 4096: 1,744 tiles on 256 CUs = 6.8 rounds) when they are launched on a LOWER-priority HIP stream?
Serial (one stream) vs two streams at equal priority vs high / low priority, one layer's backward GEMM mix repeated."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

dev = "cuda:0"
M, d, f = int(os.environ.get("M", 27664)), 4096, 11008
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16, generator=g) * 0.05
# input-gradient chain of one layer (dx = dy @ W, NN form on W itself) and the independent weight gradients (dW = dy^T x)
dy_gu, W_gu, x_d = rn(M, 2 * f), rn(2 * f, d), rn(M, d)
dy_qkv, W_qkv = rn(M, 3 * d), rn(3 * d, d)
dy_o, W_o = rn(M, d), rn(d, d)
dy_dn, W_dn, x_f = rn(M, d), rn(d, f), rn(M, f)
gW_gu, gW_qkv, gW_o, gW_dn = torch.empty_like(W_gu), torch.empty_like(W_qkv), torch.empty_like(W_o), torch.empty_like(W_dn)
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")


def dgrads():
    ops.gemm_nn(dy_dn, W_dn)
    ops.gemm_nn(dy_gu, W_gu)
    ops.gemm_nn(dy_o, W_o)
    ops.gemm_nn(dy_qkv, W_qkv)


def wgrads():
    ops.gemm_tn(dy_dn, x_f, out=gW_dn)
    ops.gemm_tn(dy_gu, x_d, out=gW_gu)
    ops.gemm_tn(dy_o, x_d, out=gW_o)
    ops.gemm_tn(dy_qkv, x_d, out=gW_qkv)


def run(mode, layers=8, reps=3):
    cur = torch.cuda.current_stream()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if mode == "serial":
            e0.record()
            for _ in range(layers):
                dgrads()
                wgrads()
            e1.record()
        else:
            hi, lo = mode
            hi.wait_stream(cur), lo.wait_stream(cur)
            with torch.cuda.stream(hi):
                e0.record()
            for _ in range(layers):
                with torch.cuda.stream(hi):
                    dgrads()
                    ev = torch.cuda.Event()
                    ev.record()
                with torch.cuda.stream(lo):
                    lo.wait_event(ev)          # the weight gradients of a layer need that layer's dy
                    wgrads()
            hi.wait_stream(lo)
            with torch.cuda.stream(hi):
                e1.record()
            cur.wait_stream(hi)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / layers)
    return best


for _ in range(2):
    dgrads(), wgrads()
print(f"M = {M}: per layer, best of 3 (8 layers each)")
print(f"  dgrads only            {run('serial') and 0 or 0:.0f}", end="\r")
torch.cuda.synchronize()
t_serial = run("serial")
print(f"  one stream (serial)               {t_serial:8.3f} ms")
s0, s1 = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=0)
print(f"  two streams, equal priority       {run((s0, s1)):8.3f} ms")
lo_p, hi_p = (torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1))
sh, sl = torch.cuda.Stream(priority=hi_p), torch.cuda.Stream(priority=lo_p)
print(f"  two streams, priority {hi_p} / {lo_p}       {run((sh, sl)):8.3f} ms   (streams report {sh.priority} / {sl.priority})")
sh2, sl2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
print(f"  two streams, priority -1 / 0      {run((sh2, sl2)):8.3f} ms   (streams report {sh2.priority} / {sl2.priority})")
print(f"  one stream again                  {run('serial'):8.3f} ms")
