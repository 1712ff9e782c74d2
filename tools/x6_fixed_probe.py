#!/usr/bin/env python3
"""Fixed cost of the persistent fp32x6 launches: time against rows, time(M) = fixed + M x per_row (least squares over M = 4096 t, t tiles per
row range), per kernel form -- and, with a variant library (tools/jobs/x6_ablation.sh, X6_ABL bits 64 .. 1024), what the fixed part is made of.
    python tools/x6_fixed_probe.py [X6_ABL bits of the variant library = 0: the shipped one]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
from contrastive_lift_amd import _lib
if v:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_scratch", "abl", f"libclift_abl{v}_s0.so")
import numpy as np
import torch
from contrastive_lift_amd import engine
dev = "cuda"
engine.set_mlp_precision("fp32x6")


def timeit(fn, n=60):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator().manual_seed(0)
W = (torch.randn(256, 256, generator=g) / 16).to(dev); b = torch.randn(256, generator=g).to(dev)
W0 = torch.randn(256, 3, generator=g).to(dev); b0 = torch.randn(256, generator=g).to(dev)
Wo = (torch.randn(3, 256, generator=g) / 16).to(dev); bo = torch.randn(3, generator=g).to(dev)
W2 = (torch.randn(128, 128, generator=g) / 12).to(dev); b2 = torch.randn(128, generator=g).to(dev)
W1 = (torch.randn(128, 160, generator=g) / 12).to(dev)
Ms = [4096 * t for t in (1, 2, 4, 8, 16, 32, 64)]
res = {}
for M in Ms:
    A = torch.relu(torch.randn(M, 256, device=dev)); dY = torch.randn(M, 256, device=dev); C_ = torch.empty(M, 256, device=dev)
    x4 = torch.cat([torch.rand(M, 3, device=dev) * 2 - 1, torch.zeros(M, 1, device=dev)], 1).contiguous()
    out = torch.empty(M, 6, device=dev); gW0 = torch.zeros(256, 3, device=dev); gb0 = torch.zeros(256, device=dev)
    gW = torch.zeros(256, 256, device=dev); gb = torch.zeros(256, device=dev)
    sb = engine.sign_bits_for(M, torch.device("cuda", 0))
    X = torch.relu(torch.randn(M, 160, device=dev)); H = torch.relu(torch.randn(M, 128, device=dev)); d1 = torch.randn(M, 128, device=dev)
    H1 = torch.empty(M, 128, device=dev); dX = torch.empty(M, 160, device=dev); g1 = torch.zeros(128, 160, device=dev); g2 = torch.zeros(128, 128, device=dev); gb1 = torch.zeros(128, device=dev)
    r = {"x6 forward": lambda: engine.gemm(M, 256, 256, A, 256, W, 256, C_, 256, bias=b, act=1),
         "x6 generated input": lambda: engine.first2_x6(M, x4, W0, b0, W, b, C_),
         "x6 output-fused": lambda: engine.last2_x6(M, A, W, b, Wo, bo, None, out, 6, 0),
         "x6 dgrad (sign bytes)": lambda: engine.gemm(M, 256, 256, dY, 256, W, 256, C_, 256, b_trans=1, sign_bits=sb),
         "x6 first2_bwd": lambda: engine.first2_x6_bwd(M, dY, W, W0, b0, x4, gW0, gb0),
         "x6 wgrad": lambda: engine.wgrad(256, 256, M, dY, 256, A, 256, gW, gb),
         "n6 forward K=160": lambda: engine.gemm(M, 128, 160, X, 160, W1, 160, H1, 128, bias=b2, act=1),
         "n6 dgrad masked": lambda: engine.gemm(M, 128, 128, d1, 128, W2, 128, H1, 128, b_trans=1, mask=H, ldmask=128),
         "n6 dgrad 160": lambda: engine.gemm(M, 160, 128, d1, 128, W1, 160, dX, 160, b_trans=1),
         "n6 wgrad 160": lambda: engine.wgrad(128, 160, M, d1, 128, X, 160, g1, gb1),
         "n6 wgrad 128": lambda: engine.wgrad(128, 128, M, d1, 128, H, 128, g2, gb1)}
    if v >= 4096:        # cache-policy variants of csrc/layer_x6.hip (X6_NT = v - 4096)
        r = {k: f for k, f in r.items() if k.startswith("x6") and "wgrad" not in k}
    elif v >= 2048:      # a variant of csrc/layer_n6.hip
        r = {k: f for k, f in r.items() if k.startswith("n6")}
    elif v:
        r = {k: f for k, f in r.items() if k in ("x6 forward", "x6 generated input", "x6 dgrad (sign bytes)")}
    for k, f in r.items():
        res.setdefault(k, []).append(timeit(f))
print(f"X6_ABL {v}: rows " + " ".join(f"{m:>7d}" for m in Ms))
for k, ts in res.items():
    a, c = np.polyfit(np.array(Ms, dtype=np.float64), np.array(ts), 1)
    print(f"  {k:24s} " + " ".join(f"{t:7.1f}" for t in ts) + f"   us  | fixed {c:6.1f} us + {a * 1e3:6.3f} ns / row")
