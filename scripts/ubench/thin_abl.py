import os, sys, torch
sys.path.insert(0, '/root/repo')
from layoutdit_amd import ops
dev='cuda:0'
os.environ["LDIT_GEMM_TILE"]="4"
for (M,N,K) in [(197,768,768),(197,3072,768),(197,768,3072)]:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.05; b=torch.randn(N,device=dev); y=torch.empty(M,N,device=dev)
    line=f"M={M} N={N} K={K}:"
    for abl in (0,1,2,3,4,8,12,15):
        os.environ["THIN_ABL"]=str(abl)
        for _ in range(5): ops.linear(x,w,b,out=y)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): ops.linear(x,w,b,out=y)
        e1.record(); torch.cuda.synchronize()
        line+=f"  abl{abl}: {e0.elapsed_time(e1)/40*1e3:6.1f}"
    print(line, flush=True)
