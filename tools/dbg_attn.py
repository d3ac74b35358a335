import os, sys, torch, math
sys.path.insert(0, os.getcwd())
from rlaif_v_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda:0")
S, H, hd, L, causal = 2, 2, 128, 97, False
g = torch.Generator().manual_seed(L)
qkv = (torch.randn(S * L, 3 * H * hd, generator=g) * 0.7).to(torch.bfloat16).to(dev)
do = torch.randn(S * L, H * hd, generator=g).to(torch.bfloat16).to(dev)
out, lse = ops.attn_fwd(qkv, S, L, H, hd, causal, 0, H * hd, 2 * H * hd)
res = []
for it in range(3):
    dqkv = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, causal, 0, H * hd, 2 * H * hd)
    torch.cuda.synchronize()
    res.append(dqkv.float().cpu())
torch.save(res, "gpurun_out/dbg_v%s.pt" % os.environ.get("RV_ATTN_DKV", "5"))
x = res[0]
bad = (~torch.isfinite(x)) | (x.abs() > 1e3)
print("v", os.environ.get("RV_ATTN_DKV"), "bad count", int(bad.sum()), "runs equal", all(torch.equal(res[0], r) for r in res))
idx = bad.nonzero()
print(idx[:20].tolist())
