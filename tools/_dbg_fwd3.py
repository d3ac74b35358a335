import sys, os, math, torch
sys.path.insert(0, '/root/repo')
from rlaif_v_amd import hip, ops
from tools.exp_attn_fwd3 import ref_attn
dev = torch.device("cuda:0"); lib = hip.lib(); BF = torch.bfloat16
S, L, H = 1, 2048, 8
hd = 128; d = H * hd
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(S * L, 3 * d, generator=g) * 0.7).to(BF).to(dev)
ro, rl = ref_attn(qkv, S, L, H, hd, True)
lib.call("rv_set_attn_fwd_version", 3)
for rep in range(3):
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, 2 * d)
    torch.cuda.synchronize()
    dl = (lse - rl).abs()
    idx = torch.nonzero(dl > 3e-3)
    print(f"rep{rep}: bad {idx.shape[0]}", "by head:", torch.bincount(idx[:, 1], minlength=H).tolist())
    x = qkv.float()
    for h in range(H):
        qs = idx[idx[:, 1] == h][:, 2].tolist()
        if not qs:
            continue
        q = x[:, h * hd:(h + 1) * hd]; k = x[:, d + h * hd:d + (h + 1) * hd]
        sc = (q @ k.t()) / math.sqrt(hd)           # [L, L] unmasked scores
        e = sc.exp()
        print(" h", h, "rows", qs[0], "..", qs[-1], "n", len(qs))
        for qq in qs[:3] + qs[-2:]:
            extra = math.exp(float(lse[0, h, qq])) - math.exp(float(rl[0, h, qq]))
            k0 = (qq // 64) * 64
            cands = {"rest of diag tile": float(e[qq, qq + 1:k0 + 64].sum()), "tile+1": float(e[qq, k0 + 64:k0 + 128].sum()),
                     "tile+2": float(e[qq, k0 + 128:k0 + 192].sum()), "tile-1": float(e[qq, k0 - 64:k0].sum()), "tile 0": float(e[qq, 0:64].sum()),
                     "diag visible": float(e[qq, k0:qq + 1].sum()), "half1 of diag (all)": float(e[qq, k0 + 32:k0 + 64].sum())}
            print(f"    q {qq}: extra mass {extra:.3f} (ref total {math.exp(float(rl[0, h, qq])):.1f});", {a: round(b, 2) for a, b in cands.items()},
                  " out err", float((out[qq, h * hd:(h + 1) * hd].float() - ro[qq, h * hd:(h + 1) * hd]).abs().max()))
lib.call("rv_set_attn_fwd_version", 0)
