#!/usr/bin/env python3
"""Launch time of the 256->256 layer GEMMs as a function of the row count M (wave-quantisation probe; run on the GPU box).
One chip-round of k_gemm<128,256> is 512 resident blocks x 128 rows = 65536 rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine

dev = "cuda"
Ms = [int(x) for x in sys.argv[1:]] or [32768, 65536, 66000, 98304, 131072, 132000, 147456, 163840, 174000, 180224, 196608, 197000, 229376, 262144, 265000]
Mmax = max(Ms)
A = torch.randn(Mmax, 256, device=dev); W = torch.randn(256, 256, device=dev); Cm = torch.zeros(Mmax, 256, device=dev)
bias = torch.randn(256, device=dev)
for M in Ms:
    row = []
    for name, kw in (("fwd", dict(bias=bias, act=1)), ("dgrad", dict(b_trans=1, mask=A, ldmask=256))):
        for _ in range(3):
            engine.gemm(M, 256, 256, A, 256, W, 256, Cm, 256, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            engine.gemm(M, 256, 256, A, 256, W, 256, Cm, 256, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append(f"{name} {us:7.1f} us {2.0 * M * 65536 / us / 1e6:6.1f} TF")
    print(f"M={M:7d} rounds={M / 65536:5.2f}  " + "   ".join(row))
