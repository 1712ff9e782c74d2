#!/usr/bin/env python3
"""Micro-soak of ONE launch: the appearance MLP's fused last two layers (128 -> 128 -> 3 + sigmoid) on the hidden activation of a real frame chunk.

    python tools/last2_soak.py <seconds per variant> [variants, comma separated: exact_infer,exact_keep,x6_infer,x6_keep,exact_unfused]

The render soak (tools/determinism_soak.py, profiles/r05_determinism.txt) names this launch as the only one of the exact-fp32 mode whose output was
ever seen to differ between two renders of the same rays in ONE process (1 of 8000 renders = 16000 launches; its input H1 bit-identical, its
output rgb_s different in 17 rows by ~1e-6).  Here the launch runs alone, back to back, on the same input, and every output is compared bit for
bit with the first: `*_infer` = the frame-render form (hidden activation not written), `*_keep` = the training form (H2 written: tells whether the
128 x 128 product or the output layer went wrong), `exact` = clift_app_head_last2_fwd (csrc/layer_n128.hip), `x6` = clift_app_head_last2_x6_fwd
(csrc/layer_n6.hip), `exact_unfused` = the three-launch form (layer, output layer, sigmoid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastive_lift_amd import engine, synthetic
from contrastive_lift_amd._lib import call, ptr, stream

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
variants = (sys.argv[2] if len(sys.argv) > 2 else "exact_infer,exact_keep,x6_infer,x6_keep").split(",")
dev = torch.device("cuda:0")
engine.set_mlp_precision("fp32")
engine.SOAK_KEEP = True
model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device=dev)
renderer.update_step_ratio(renderer.step_ratio * 0.5)
with torch.no_grad():
    o, ctx = engine.render_forward(model, renderer, pool[:65536].contiguous(), None, False, grad_heads=())
H1 = ctx.soak["H1"].clone()
M = H1.shape[0]
del o, ctx
torch.cuda.empty_cache()
views = model.named_views()
W2, b2 = views["render_appearance_mlp.mlp.2.weight"], views["render_appearance_mlp.mlp.2.bias"]
W3, b3 = views["render_appearance_mlp.mlp.4.weight"], views["render_appearance_mlp.mlp.4.bias"]
print(f"M = {M} rows, H1 {tuple(H1.shape)}, |H1| max {float(H1.abs().max()):.3g}, nonzero {float((H1 != 0).float().mean()):.3f}", flush=True)


def launch(kind, H2, rgb):
    if kind == "exact":
        engine.app_last2(M, H1, W2, b2, W3, b3, H2, rgb)
    elif kind == "x6":
        call("clift_app_head_last2_x6_fwd", ptr(H1), 128, ptr(W2), engine._pitch(W2), ptr(b2), ptr(W3), engine._pitch(W3), ptr(b3), 3, M, ptr(H2), 128,
             ptr(rgb), 3, 1, stream())
    else:           # exact_unfused: 128 x 128 layer, 3-wide output layer, row sigmoid
        with engine.exact_fp32():
            engine.gemm(M, 128, 128, H1, 128, W2, engine._pitch(W2), H2, 128, bias=b2, act=1)
            pre = torch.empty((M, 3), dtype=torch.float32, device=dev)
            engine.gemm(M, 3, 128, H2, 128, W3, engine._pitch(W3), pre, 3, bias=b3)
            call("clift_rows_act_fwd", ptr(pre), 3, M, 3, 1, ptr(rgb), 3, stream())


def describe(name, a, b):
    d = a.view(torch.int32) != b.view(torch.int32)
    idx = torch.nonzero(d)
    rows = torch.unique(idx[:, 0])
    cols = torch.unique(idx[:, 1])
    r0 = int(rows[0])
    first = idx[0].tolist()
    return (f"{name}: {int(d.sum())} elements in {rows.numel()} rows [{r0} .. {int(rows[-1])}] (row % 64: {sorted(set((rows % 64).tolist()))[:40]}), "
            f"columns {cols.tolist()[:40]}; first {first} got {float(a[tuple(first)])!r} want {float(b[tuple(first)])!r}; max |delta| {float((a - b).abs().max()):.3g}")


for v in variants:
    kind, form = v.rsplit("_", 1) if not v.endswith("unfused") else ("unfused", "keep")
    keep = form == "keep"
    H2 = torch.empty((M, 128), dtype=torch.float32, device=dev) if keep else None
    rgb = torch.empty((M, 3), dtype=torch.float32, device=dev)
    launch(kind, H2, rgb)
    ref_rgb, ref_H2 = rgb.clone(), (H2.clone() if keep else None)
    torch.cuda.synchronize()
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        for _ in range(20):
            rgb.fill_(-1.0)
            launch(kind, H2, rgb)
            eq = torch.equal(rgb, ref_rgb) and (not keep or torch.equal(H2, ref_H2))
            n += 1
            if not eq:
                bad += 1
                if bad <= 8:
                    if not torch.equal(rgb, ref_rgb):
                        print(v, "launch", n, describe("rgb", rgb, ref_rgb), flush=True)
                    if keep and not torch.equal(H2, ref_H2):
                        print(v, "launch", n, describe("H2", H2, ref_H2), flush=True)
    torch.cuda.synchronize()
    print(f"LAST2 RESULT {v}: {bad} of {n} launches differ from the first ({time.time() - t0:.0f} s, {1e3 * (time.time() - t0) / n:.2f} ms per launch incl. compare)", flush=True)
