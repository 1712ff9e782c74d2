import sys, os
sys.path.insert(0, "/root/repo")
import torch
from contrastive_lift_amd import engine
dev="cuda"
for (M,N,K) in [(262144,256,256),(65536,256,1024),(16384,256,4096),(262144,256,64),(262144,256,128),(262144,256,512), (4096,4096,4096)]:
    A=torch.randn(M,K,device=dev); B=torch.randn(N,K,device=dev); C=torch.empty(M,N,device=dev)
    for _ in range(3): engine.gemm(M,N,K,A,K,B,K,C,N)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): engine.gemm(M,N,K,A,K,B,K,C,N)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF")
