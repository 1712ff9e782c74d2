#!/usr/bin/env python3
"""Does it pay to run a forward-only xyz head (K = 3 layer + 256 x 256 layers + narrow output, fp32x6) over ROW BLOCKS small enough for the hidden
activation between two launches to stay in the 256 MB memory-side cache, instead of streaming the whole chunk (M ~ 5 - 9 M rows at frame-render
chunk sizes: ~5 - 9 GB per hidden tensor) through HBM launch by launch?  Same launches, same arithmetic, same result bits; only the order changes.
    python tools/mall_block_probe.py [M total = 4194304]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine
dev = torch.device("cuda", 0)
engine.set_mlp_precision("fp32x6")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) / 16).to(dev)
W0, b0 = torch.randn(256, 3, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
layers5 = [(W0, b0), (mk(256, 256), mk(256)), (mk(256, 256), mk(256)), (mk(256, 256), mk(256)), (mk(22, 256), mk(22))]     # semantic head (C = 22)
layers4 = [(W0, b0), (mk(256, 256), mk(256)), (mk(256, 256), mk(256)), (mk(3, 256), mk(3))]                                 # instance head (E = 3)
xa = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).contiguous().to(dev)


def run(layers, out, block):
    E = layers[-1][0].shape[0]
    for r0 in range(0, M, block):
        r1 = min(M, r0 + block)
        engine.xyz_mlp_fwd(layers, xa[r0:r1], r1 - r0, out[r0:r1], out.shape[1], 0, keep_first=False, out_act=2 if E == 22 else 0)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, layers in (("semantic head (3 split launches + output layer)", layers5), ("instance head (2 split launches)", layers4)):
    E = layers[-1][0].shape[0]
    ref = torch.empty((M, E), device=dev)
    run(layers, ref, M)
    base = timed(lambda: run(layers, ref, M))
    print(f"{name}, M = {M}: whole chunk per launch {base:8.2f} ms ({base * 1e6 / M:.3f} ns / row)", flush=True)
    for block in (1048576, 524288, 262144, 131072, 65536, 32768):
        out = torch.empty((M, E), device=dev)
        t = timed(lambda: run(layers, out, block))
        print(f"    row blocks of {block:8d}: {t:8.2f} ms ({t * 1e6 / M:.3f} ns / row, x {base / t:.3f}), bit-identical to the whole-chunk result: {torch.equal(out, ref)}", flush=True)
