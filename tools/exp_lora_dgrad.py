"""LoRA input gradient under adapter dropout: one pass (rv_gemm_nn_lora_pre_bf16) against plain NN GEMM + rv_gemm_nt_dropout_bf16,
on the four projection shapes of a 7B decoder layer (config 5: L = 4096, 4 pairs).  Usage: python tools/exp_lora_dgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=8, warmup=2):
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
    dev = torch.device("cuda:0")
    M = int(os.environ.get("ROWS", "29000"))
    g = torch.Generator(device=dev).manual_seed(0)
    tot = {"0": 0.0, "1": 0.0}
    for name, out_w, in_w, G in (("qkv", 12288, 4096, 3), ("o", 4096, 4096, 1), ("gate|up", 22016, 4096, 2), ("down", 4096, 11008, 1)):
        dy = torch.randn(M, out_w, device=dev, generator=g).to(BF)
        w = (torch.randn(out_w, in_w, device=dev, generator=g) * 0.02).to(BF)
        wT = w.t().contiguous()
        dt = torch.randn(M, 64 * G, device=dev, generator=g).to(BF)
        a = (torch.randn(64 * G, in_w, device=dev, generator=g) * 0.1).to(BF)
        aT = a.t().contiguous()
        row = []
        for mode in ("0", "1", "0", "1"):
            os.environ["RV_LORA_DGRAD_PRE"] = mode
            t = timeit(lambda: ops.lora_dgrad_dropout(dy, w, wT, dt, a, aT, 0.05, 17))
            row.append(t)
            tot[mode] += t / 2
        plain = timeit(lambda: ops.linear(dy, wT, w))
        print(f"{name:8s} K={out_w:6d} N={in_w:6d} K2={64 * G:4d}: two kernels {row[0]:.3f} / {row[2]:.3f} ms, one pass {row[1]:.3f} / "
              f"{row[3]:.3f} ms, plain input gradient alone {plain:.3f} ms", flush=True)
        del dy, w, wT, dt, a, aT
    print(f"per layer: two kernels {tot['0']:.3f} ms, one pass {tot['1']:.3f} ms -> {32 * (tot['0'] - tot['1']):.1f} ms per 32-layer step")


if __name__ == "__main__":
    main()
