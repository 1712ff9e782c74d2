#!/usr/bin/env python3
"""fp32x6 persistent layer kernels (csrc/layer_x6.hip) on the GPU box: error against fp64 next to the exact-fp32 kernel's, and launch times
against the exact persistent kernel and the tiled split kernel.   tools/x6_probe.py [M ...]"""
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


def run(mode, fn):
    prev = engine.set_mlp_precision(mode)
    try:
        return fn()
    finally:
        engine.set_mlp_precision(prev)


g = torch.Generator().manual_seed(3)
for M in (1, 31, 300, 1000, 4097, 66001):
    A = (torch.relu(torch.randn(M, 256, generator=g)) * torch.exp(1.5 * torch.randn(M, 1, generator=g))).to(dev)
    W = (torch.randn(256, 256, generator=g) / 16).to(dev); b = torch.randn(256, generator=g).to(dev)
    mk = torch.randn(M, 256, generator=g).to(dev)
    ref_f = torch.relu(A.double() @ W.double().T + b.double())
    ref_d = (A.double() @ W.double()) * (mk > 0)
    out = {}
    for mode in ("fp32", "fp32x6"):
        def both():
            C1 = torch.full((M, 260), -7.0, device=dev); C2 = torch.full((M, 256), -7.0, device=dev)
            engine.gemm(M, 256, 256, A, 256, W, 256, C1, 260, bias=b, act=1)
            engine.gemm(M, 256, 256, A, 256, W, 256, C2, 256, b_trans=1, mask=mk, ldmask=256)
            return C1, C2
        out[mode] = run(mode, both)
    sc_f = ref_f.abs().amax(1, keepdim=True).clamp_min(1e-30); sc_d = ref_d.abs().amax(1, keepdim=True).clamp_min(1e-30)
    e = {m: (float(((o[0][:, :256].double() - ref_f).abs() / sc_f).max()), float(((o[1].double() - ref_d).abs() / sc_d).max())) for m, o in out.items()}
    pad_ok = bool((out["fp32x6"][0][:, 256:] == -7.0).all())
    print(f"M={M:6d}  fwd err/rowmax fp32 {e['fp32'][0]:.2e} x6 {e['fp32x6'][0]:.2e}   dgrad fp32 {e['fp32'][1]:.2e} x6 {e['fp32x6'][1]:.2e}   pad untouched {pad_ok}")

for M in [int(x) for x in sys.argv[1:]] or [249000, 62000]:
    A = torch.relu(torch.randn(M, 256, device=dev)); W = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev)
    mk = torch.relu(torch.randn(M, 256, device=dev)); C_ = torch.empty(M, 256, device=dev); dY = torch.randn(M, 256, device=dev)
    fwd = lambda: engine.gemm(M, 256, 256, A, 256, W, 256, C_, 256, bias=b, act=1)
    dgr = lambda: engine.gemm(M, 256, 256, dY, 256, W, 256, C_, 256, b_trans=1, mask=mk, ldmask=256)
    fl = 2.0 * M * 65536
    for name, f in (("fwd", fwd), ("dgrad", dgr)):
        t32 = run("fp32", lambda: timeit(f))
        tx6 = run("fp32x6", lambda: timeit(f))
        os.environ["CLIFT_X6_TILED"] = "1"
        txt = run("fp32x6", lambda: timeit(f))
        del os.environ["CLIFT_X6_TILED"]
        print(f"{name} 256x256 M={M}: exact persistent {t32:7.1f} us ({fl/t32/1e6:6.1f} TF)   x6 persistent {tx6:7.1f} us ({fl/tx6/1e6:6.1f} TF-equiv, "
              f"{2*4*M*256/tx6/1e3:5.0f} GB/s)   x6 tiled {txt:7.1f} us")
