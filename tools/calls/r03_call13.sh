#!/bin/bash
# Round-3 GPU call 13: the rewritten colsum kernel - parity tests that reach it + a timing at the projector's shape.
set -u
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_omnilmm_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from rlaif_v_amd import ops
x = torch.randn(4608, 4096, device='cuda').bfloat16()
for _ in range(3): ops.colsum(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): y = ops.colsum(x)
e.record(); torch.cuda.synchronize()
print("colsum 4608 x 4096:", s.elapsed_time(e) / 20 * 1e3, "us; max err", (y.float() - x.float().sum(0)).abs().max().item())
PY
