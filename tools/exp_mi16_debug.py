"""Debug aid for the 16x16x32-MFMA NN GEMM: C = I * B with B[k][n] = k and B[k][n] = n reveals which (k, n) source every
output element received."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("rlaif-v_amd.ops")
M, N, K = 256, 256, 512
dev = "cuda"
A = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
A[torch.arange(M), torch.arange(M)] = 1
Bk = torch.arange(K, device=dev, dtype=torch.float32)[:, None].expand(K, N).clamp(max=255).to(torch.bfloat16).contiguous()
Bn = torch.arange(N, device=dev, dtype=torch.float32)[None, :].expand(K, N).to(torch.bfloat16).contiguous()
Ck = ops.gemm_nn(A, Bk).float().cpu()
Cn = ops.gemm_nn(A, Bn).float().cpu()
m = torch.arange(M)[:, None].expand(M, N).float()
n = torch.arange(N)[None, :].expand(M, N).float()
print("rows wrong:", int((Ck != m).sum()), "cols wrong:", int((Cn != n).sum()), "of", M * N)
bad = torch.nonzero((Ck != m) | (Cn != n))
for i in range(min(40, bad.shape[0])):
    r, c = int(bad[i, 0]), int(bad[i, 1])
    print(f"out[{r}][{c}] got (k={int(Ck[r, c])}, n={int(Cn[r, c])})")
# structure summary
dk = (Ck - m).flatten().tolist(); dn = (Cn - n).flatten().tolist()
from collections import Counter
print("k deltas:", Counter(dk).most_common(12))
print("n deltas:", Counter(dn).most_common(12))
