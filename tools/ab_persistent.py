#!/usr/bin/env python3
"""A/B on the GPU box: gradients of the mid-size parity scene with the persistent kernels (default) vs the tiled kernels
(CLIFT_NO_PERSISTENT=1).  Same inputs; prints per tensor the largest difference relative to the tensor's scale and how many entries
differ by more than 1e-4 of it -- a handful of ReLU-kink / activity-threshold flips is expected, anything systematic is a bug."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from test_gpu_parity import _import, build_model, scene, _run_forward_backward
cl, op, orender, ofld, olosses, orays = _import()
res, C_, E, n_rays = (40, 48, 56), 22, 3, 900
aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
P, rays, rng = scene(op, orays, 23, res, C_, E, n_rays, amp=2.2, sg=0.4)
jitter = torch.from_numpy(rng.uniform(0, 1, n_rays).astype(np.float32))
cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((n_rays, 3), (n_rays, C_), (n_rays, 2 * E))]
out = {}
for tag in ("persistent", "tiled"):
    if tag == "tiled":
        os.environ["CLIFT_NO_PERSISTENT"] = "1"
    m = build_model(cl, P, res, C_, E, -3.0, "softmax")
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to("cuda")
    outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    out[tag] = (outs, {k: (None if g is None else g.detach().double().cpu()) for k, g in grads.items()})
for i, nm in enumerate(("rgb", "sem", "inst")):
    a, b = out["persistent"][0][i].double().cpu(), out["tiled"][0][i].double().cpu()
    print(f"{nm:40s} max |diff| / scale {float((a - b).abs().max() / b.abs().max()):.3e}")
for k, b in out["tiled"][1].items():
    a = out["persistent"][1][k]
    if a is None or b is None:
        continue
    sc = float(b.abs().max())
    d = (a - b).abs()
    print(f"{k:40s} max |diff| / scale {float(d.max()) / max(sc, 1e-30):.3e}   entries > 1e-4 scale: {int((d > 1e-4 * sc).sum())}/{d.numel()}")
