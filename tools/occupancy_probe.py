import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine
dev = "cuda"
N = K = 256
for blocks in (128, 256, 384, 512, 768, 1024, 2048):
    M = 128 * blocks
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    for _ in range(3): engine.gemm(M, N, K, A, K, B, K, C, N)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): engine.gemm(M, N, K, A, K, B, K, C, N)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"blocks={blocks:5d}  {ms*1e3:7.1f} us   {2.0*M*N*K/ms/1e9:6.1f} TF")
