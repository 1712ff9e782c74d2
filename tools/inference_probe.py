#!/usr/bin/env python3
"""Frame-render probe (BASELINE configs[4] shape): renders a 262144-ray tile with is_train=False and prints rays/s.
Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contrastive_lift_amd as cl
from contrastive_lift_amd import synthetic, inference as inf, engine

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
engine.set_mlp_precision(dtype)
model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device="cuda")
renderer.update_step_ratio(renderer.step_ratio * 0.5)
rays = pool[:262144].contiguous()
inf.render_rays(model, renderer, rays[:chunk], chunk)
torch.cuda.synchronize(); t = time.perf_counter()
inf.render_rays(model, renderer, rays, chunk)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"{dtype} chunk {chunk}: {rays.shape[0]/dt:,.0f} rays/s, {rays.shape[0]*renderer.n_samples/dt/1e6:,.1f} M ray-samples/s, S={renderer.n_samples}")
