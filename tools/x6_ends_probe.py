#!/usr/bin/env python3
"""Launch times of every fp32x6 kernel form next to the exact-fp32 form it replaces (20 launches back to back, torch events):
plain forward / GEN / OUTV with and without the hidden store / masked dgrad / K3W / weight gradient / generated-input weight gradient.
    python tools/x6_ends_probe.py [M ...]"""
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


def mode(m, fn):
    prev = engine.set_mlp_precision(m)
    try:
        return fn()
    finally:
        engine.set_mlp_precision(prev)


for M in [int(x) for x in sys.argv[1:]] or [249000, 62000]:
    g = torch.Generator().manual_seed(1)
    A = torch.relu(torch.randn(M, 256, device=dev)); W = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev)
    W0 = torch.randn(256, 3, device=dev); b0 = torch.randn(256, device=dev)
    Wo = torch.randn(3, 256, device=dev) / 16; bo = torch.randn(3, device=dev)
    x4 = torch.cat([torch.rand(M, 3, device=dev) * 2 - 1, torch.zeros(M, 1, device=dev)], 1).contiguous()
    mk = torch.relu(torch.randn(M, 256, device=dev)); C_ = torch.empty(M, 256, device=dev); dY = torch.randn(M, 256, device=dev)
    out = torch.empty(M, 6, device=dev); gW = torch.zeros(256, 256, device=dev); gb = torch.zeros(256, device=dev)
    gW0 = torch.zeros(256, 3, device=dev); gb0 = torch.zeros(256, device=dev)
    rows = []
    t = lambda m, f: mode(m, lambda: timeit(f))
    rows.append(("forward plain", t("fp32", lambda: engine.gemm(M, 256, 256, A, 256, W, 256, C_, 256, bias=b, act=1)),
                 t("fp32x6", lambda: engine.gemm(M, 256, 256, A, 256, W, 256, C_, 256, bias=b, act=1))))
    rows.append(("forward GEN (K=3 layer generated)", timeit(lambda: engine.first2(M, x4, W0, b0, W, b, None, C_)),
                 timeit(lambda: engine.first2_x6(M, x4, W0, b0, W, b, C_))))
    rows.append(("forward OUTV, hidden kept", timeit(lambda: engine.last2(M, A, W, b, Wo, bo, C_, out, 6, 0)),
                 timeit(lambda: engine.last2_x6(M, A, W, b, Wo, bo, C_, out, 6, 0))))
    rows.append(("forward OUTV, hidden not written", timeit(lambda: engine.last2(M, A, W, b, Wo, bo, None, out, 6, 0)),
                 timeit(lambda: engine.last2_x6(M, A, W, b, Wo, bo, None, out, 6, 0))))
    rows.append(("masked dgrad", t("fp32", lambda: engine.gemm(M, 256, 256, dY, 256, W, 256, C_, 256, b_trans=1, mask=mk, ldmask=256)),
                 t("fp32x6", lambda: engine.gemm(M, 256, 256, dY, 256, W, 256, C_, 256, b_trans=1, mask=mk, ldmask=256))))
    rows.append(("first2_bwd (K3W)", timeit(lambda: engine.first2_bwd(M, dY, W, W0, b0, x4, gW0, gb0)),
                 timeit(lambda: engine.first2_x6_bwd(M, dY, W, W0, b0, x4, gW0, gb0))))
    rows.append(("weight gradient 256x256", t("fp32", lambda: engine.wgrad(256, 256, M, dY, 256, A, 256, gW, gb)),
                 t("fp32x6", lambda: engine.wgrad(256, 256, M, dY, 256, A, 256, gW, gb))))
    rows.append(("first2_wgrad (generated X)", timeit(lambda: engine.first2_wgrad(M, dY, W0, b0, x4, gW, gb)),
                 timeit(lambda: engine.first2_x6_wgrad(M, dY, W0, b0, x4, gW, gb))))
    print(f"M = {M}")
    for nm, a, c in rows:
        print(f"  {nm:36s} exact {a:7.1f} us   fp32x6 {c:7.1f} us   ({a / c:4.2f} x)")
