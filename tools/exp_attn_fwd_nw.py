#!/usr/bin/env python
"""A/B of the forward attention kernel's workgroup width at the bench shape (8 packed pair rows x 3,458 tokens, 32 heads):
RV_ATTN_FWD_NW=4 (128-query workgroups, two per CU) vs 8 (256-query workgroups sharing one K / V ring).  Each variant runs in its
own process (the knob is read once), three timing rounds each (the first round of a process runs at a ramping clock); prints the
SHA-256 of the output so that the two variants can be seen to agree bit for bit.  Also L = 4096 plain causal rows (config 5).
Usage: python tools/exp_attn_fwd_nw.py            (on the GPU box)"""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from rlaif_v_amd import ops
    from tools.bench_hot_kernels import packed_attention_inputs, timeit
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    qkv, do, seg, L, d = packed_attention_inputs(dev)
    H, hd, B = 32, 128, 8
    out, lse = ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, seg=seg)
    sha = hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes() + lse.cpu().numpy().tobytes()).hexdigest()[:16]
    ms = [timeit(lambda: ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, seg=seg, out=out), 20) for _ in range(3)]
    L2 = 4096
    q2 = (torch.randn(4 * L2, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
    o2, l2 = ops.attn_fwd(q2, 4, L2, H, hd, True, 0, d, 2 * d)
    sha2 = hashlib.sha256(o2.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    ms2 = [timeit(lambda: ops.attn_fwd(q2, 4, L2, H, hd, True, 0, d, 2 * d, out=o2), 20) for _ in range(3)]
    print(f"NW={os.environ.get('RV_ATTN_FWD_NW', 'default')}: packed 8 x {L}: {' '.join(f'{m:.3f}' for m in ms)} ms  sha {sha} | "
          f"plain 4 x {L2}: {' '.join(f'{m:.3f}' for m in ms2)} ms  sha {sha2}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rnd in range(2):
            for nw in ("4", "8"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, RV_ATTN_FWD_NW=nw))
