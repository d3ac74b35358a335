import torch, time
torch.manual_seed(0)
def t(M,N,K,tn=False):
    a=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); b=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)
    if tn:
        a=torch.randn(K,M,device='cuda',dtype=torch.bfloat16); b=torch.randn(K,N,device='cuda',dtype=torch.bfloat16)
        f=lambda: a.t()@b
    else:
        f=lambda: a@b.t()
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/10
    print(f"hipblaslt M={M} N={N} K={K} tn={tn}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s")
R=27664
for (M,N,K) in [(R,12288,4096),(R,4096,4096),(R,22016,4096),(R,4096,11008),(R,32000,4096)]: t(M,N,K)
for (M,N,K) in [(4096,12288,R),(11008,4096,R),(4096,22016,R)]: t(M,N,K,True)
