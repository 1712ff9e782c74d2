#!/usr/bin/env python3
"""Launch times of the last two layers of an instance head (256x256 hidden + E = 3 output): exact fused kernel, fp32x6 fused kernel, and the
fp32x6 layer followed by the narrow output GEMM.   tools/x6_outv_probe.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine
dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in [int(x) for x in sys.argv[1:]] or [249000, 62000]:
    A = torch.relu(torch.randn(M, 256, device=dev)); W = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev)
    Wo = torch.randn(3, 256, device=dev) / 10; bo = torch.randn(3, device=dev)
    hid = torch.empty(M, 256, device=dev); out = torch.empty(M, 6, device=dev)
    for keep in (True, False):
        h = hid if keep else None
        t_exact = timeit(lambda: engine.last2(M, A, W, b, Wo, bo, h, out, 6, 0))
        t_x6 = timeit(lambda: engine.last2_x6(M, A, W, b, Wo, bo, h, out, 6, 0))
        def two():
            with engine._Precision(2):
                engine.gemm(M, 256, 256, A, 256, W, 256, hid, 256, bias=b, act=1)
            engine.gemm(M, 3, 256, hid, 256, Wo, 256, out, 6, bias=bo)
        t_two = timeit(two)
        print(f"M={M} hidden {'kept' if keep else 'dropped'}: exact fused {t_exact:7.1f} us   fp32x6 fused {t_x6:7.1f} us   fp32x6 layer + output GEMM {t_two:7.1f} us")
