#!/usr/bin/env python3
"""Times k_layer_x6 (plain forward, output-fused without hidden store, first-two-layers backward) of one ablated library variant
(tools/jobs/x6_ablation.sh).   python tools/x6_ablation.py <X6_ABL bits>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
v, sg = int(sys.argv[1]), int(sys.argv[2])
from contrastive_lift_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_scratch", "abl", f"libclift_abl{v}_s{sg}.so")
import torch
from contrastive_lift_amd import engine
dev = "cuda"
engine.set_mlp_precision("fp32x6")


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 249000
A = torch.relu(torch.randn(M, 256, device=dev)); W = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev)
W0 = torch.randn(256, 3, device=dev); b0 = torch.randn(256, device=dev); Wo = torch.randn(3, 256, device=dev) / 16; bo = torch.randn(3, device=dev)
x4 = torch.cat([torch.rand(M, 3, device=dev) * 2 - 1, torch.zeros(M, 1, device=dev)], 1).contiguous()
C_ = torch.empty(M, 256, device=dev); dY = torch.randn(M, 256, device=dev); out = torch.empty(M, 6, device=dev)
gW0 = torch.zeros(256, 3, device=dev); gb0 = torch.zeros(256, device=dev)
names = {1: "frag reads", 2: "split", 4: "exchange", 8: "DMA", 16: "stores", 32: "barrier"}
what = " + ".join(n for bit, n in names.items() if v & bit) or "nothing"
t_f = timeit(lambda: engine.gemm(M, 256, 256, A, 256, W, 256, C_, 256, bias=b, act=1))
t_g = timeit(lambda: engine.first2_x6(M, x4, W0, b0, W, b, C_))
t_k = timeit(lambda: engine.first2_x6_bwd(M, dY, W, W0, b0, x4, gW0, gb0))
mk = torch.relu(torch.randn(M, 256, device=dev))
t_d = timeit(lambda: engine.gemm(M, 256, 256, dY, 256, W, 256, C_, 256, b_trans=1, mask=mk, ldmask=256))
t_o = timeit(lambda: engine.last2_x6(M, A, W, b, Wo, bo, None, out, 6, 0))
print(f"ABL {v:2d} stagger {sg} (without {what:22s}): forward {t_f:6.1f}  generated input {t_g:6.1f}  output-fused {t_o:6.1f}  dgrad {t_d:6.1f}  first2_bwd {t_k:6.1f} us", flush=True)
