#!/usr/bin/env python3
"""Launch times of the persistent layer kernels vs the tiled kernel (CLIFT_NO_PERSISTENT=1), on the GPU box:
   tools/layer_probe.py [M]   -- 128-wide forward K=160 / K=128, 128-wide dgrad, fused first-two-layers forward of an xyz head."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine
from contrastive_lift_amd._lib import call, ptr, stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 249000
dev = "cuda"

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for K in (160, 128):
    A = torch.relu(torch.randn(M, K, device=dev)); W = torch.randn(128, K, device=dev) / 12; b = torch.randn(128, device=dev); C_ = torch.empty(M, 128, device=dev)
    f = lambda: engine.gemm(M, 128, K, A, K, W, K, C_, 128, bias=b, act=1)
    t = timeit(f); os.environ["CLIFT_NO_PERSISTENT"] = "1"; t0 = timeit(f); del os.environ["CLIFT_NO_PERSISTENT"]
    print(f"fwd 128x{K} M={M}: persistent {t:7.1f} us ({2.0*M*128*K/t/1e6:6.1f} TF)   tiled {t0:7.1f} us ({2.0*M*128*K/t0/1e6:6.1f} TF)")
A = torch.randn(M, 128, device=dev); W = torch.randn(128, 128, device=dev) / 12; mk = torch.randn(M, 128, device=dev); C_ = torch.empty(M, 128, device=dev)
f = lambda: engine.gemm(M, 128, 128, A, 128, W, 128, C_, 128, b_trans=1, mask=mk, ldmask=128)
t = timeit(f); os.environ["CLIFT_NO_PERSISTENT"] = "1"; t0 = timeit(f); del os.environ["CLIFT_NO_PERSISTENT"]
print(f"dgrad 128x128 M={M}: persistent {t:7.1f} us ({2.0*M*128*128/t/1e6:6.1f} TF)   tiled {t0:7.1f} us ({2.0*M*128*128/t0/1e6:6.1f} TF)")
xa = torch.rand(M, 4, device=dev) * 2 - 1; W0 = torch.randn(256, 4, device=dev); b0 = torch.randn(256, device=dev)
W1 = torch.randn(256, 256, device=dev) / 16; b1 = torch.randn(256, device=dev)
h1 = torch.empty(M, 256, device=dev); h2 = torch.empty(M, 256, device=dev)
def two():
    call("clift_linear_k3_fwd", ptr(xa), ptr(W0), 4, ptr(b0), M, 256, 1, ptr(h1), 256, 0, stream())
    engine.gemm(M, 256, 256, h1, 256, W1, 256, h2, 256, bias=b1, act=1)
t_two = timeit(two)
t_keep = timeit(lambda: call("clift_xyz_head_first2_fwd", ptr(xa), ptr(W0), 4, ptr(b0), ptr(W1), 256, ptr(b1), M, ptr(h1), 256, ptr(h2), 256, stream()))
t_drop = timeit(lambda: call("clift_xyz_head_first2_fwd", ptr(xa), ptr(W0), 4, ptr(b0), ptr(W1), 256, ptr(b1), M, None, 256, ptr(h2), 256, stream()))
print(f"xyz head first two layers M={M}: k3 + layer {t_two:7.1f} us   fused (h1 kept) {t_keep:7.1f} us   fused (h1 dropped) {t_drop:7.1f} us")
for K in (128, 160):
    dY = torch.randn(M, 128, device=dev); X = torch.randn(M, K, device=dev); gW = torch.zeros(128, K, device=dev); gb = torch.zeros(128, device=dev)
    f = lambda: engine.wgrad(128, K, M, dY, 128, X, K, gW, gb)
    t = timeit(f); os.environ["CLIFT_NO_PERSISTENT"] = "1"; t0 = timeit(f); del os.environ["CLIFT_NO_PERSISTENT"]
    print(f"wgrad 128x{K} M={M}: persistent {t:7.1f} us ({2.0*M*128*K/t/1e6:6.1f} TF)   tiled {t0:7.1f} us ({2.0*M*128*K/t0/1e6:6.1f} TF)")
for Mw in sorted({M, 62000}):
    dY = torch.randn(Mw, 256, device=dev); X = torch.relu(torch.randn(Mw, 256, device=dev)); gW = torch.zeros(256, 256, device=dev); gb = torch.zeros(256, device=dev)
    f = lambda: engine.wgrad(256, 256, Mw, dY, 256, X, 256, gW, gb)
    res = {}
    for mode in ("tiled", "slices", "quads"):
        os.environ["CLIFT_WGRAD256"] = mode
        res[mode] = timeit(f)
    del os.environ["CLIFT_WGRAD256"]
    print(f"wgrad 256x256 M={Mw}: " + "   ".join(f"{k} {v:7.1f} us ({2.0*Mw*65536/v/1e6:6.1f} TF)" for k, v in res.items()))
    # correctness of the quadrant form against the tiled one
    os.environ["CLIFT_WGRAD256"] = "quads"; gW.zero_(); gb.zero_(); f(); a = gW.clone(); ab = gb.clone()
    os.environ["CLIFT_WGRAD256"] = "tiled"; gW.zero_(); gb.zero_(); f(); del os.environ["CLIFT_WGRAD256"]
    print("   quads vs tiled: max rel diff", float((a - gW).abs().max() / gW.abs().max()), "bias", float((ab - gb).abs().max() / gb.abs().max()))
