#!/usr/bin/env python3
"""Host-side cost of a training step: run it at a tiny ray count (GPU work negligible) and profile the Python side."""
import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import synthetic
from contrastive_lift_amd.trainer import HotPathTrainer, default_config
dev = "cuda"
model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device=dev)
tr = HotPathTrainer(model, renderer, default_config(chunk=0, instance_optimization_epoch=0, late_semantic_optimization=0), current_epoch=4)
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = synthetic.make_batches(pool, rays, max(16, rays // 4), 22, 25, seed=1, device=dev)
for _ in range(5):
    tr.training_step(b)
torch.cuda.synchronize()
t = time.perf_counter()
n = 100
for _ in range(n):
    tr.training_step(b)
torch.cuda.synchronize()
print(f"rays={rays}: {(time.perf_counter() - t) / n * 1e3:.3f} ms per step (wall)")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    tr.training_step(b)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
