"""rmsnorm_bwd micro-benchmark at the 7B step shape.  Round-1 note: a wave-per-row variant (row kept in registers, DPP
reductions only) measured 546 us vs 290 us for the block-per-row kernel - 260 VGPRs leave one wave per SIMD and too few bytes
in flight (16 MB chip-wide vs ~56 MB needed at 8 TB/s x 7 us); the block-per-row kernel stays."""
import torch, sys
sys.path.insert(0,'.')
from rlaif_v_amd import ops
BF=torch.bfloat16
rows,d=27664,4096
x=torch.randn(rows,d,device='cuda').to(BF); dy=torch.randn(rows,d,device='cuda').to(BF); dres=torch.randn(rows,d,device='cuda').to(BF)
w=torch.ones(d,device='cuda',dtype=BF); dw=torch.zeros(d,device='cuda',dtype=BF)
y,rstd=ops.rmsnorm_fwd(x,w,1e-5)
for _ in range(3): ops.rmsnorm_bwd(dy,x,w,rstd,dw,dres=dres)
torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
for _ in range(20): ops.rmsnorm_bwd(dy,x,w,rstd,dw,dres=dres)
e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/20
print(f"rmsnorm_bwd {rows}x{d}: {ms*1e3:.1f} us  {4*rows*d*2/ms/1e6:.0f} GB/s")
