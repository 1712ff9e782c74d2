"""Render-level two-processes-on-one-device stress (profiles/r03_x6_notes.txt, last section): re-render the same 131 072 rays of the 128^3
bench scene `iters` times in the given MLP mode and compare every output bit for bit with the first render.  Run TWO instances at once:
    python tools/shared_gpu_render_stress.py A fp32x6 300 &  python tools/shared_gpu_render_stress.py B fp32x6 300
(modes: fp32 | bf16 | fp32x6).  One instance alone is the control."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine, synthetic, inference as inf
tag, mode, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
dev = torch.device("cuda:0")
engine.set_mlp_precision(mode)
model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device=dev)
renderer.update_step_ratio(renderer.step_ratio * 0.5)
rays = pool[:131072].contiguous()
names = ("rgb", "sem", "inst", "dist")
def run():
    o = inf.render_rays(model, renderer, rays, 65536)
    return [x.clone() for x in o[:4]]
ref = run(); torch.cuda.synchronize()
bad = {n: 0 for n in names}
t0 = time.time()
for it in range(iters):
    o = run()
    for n, a, b in zip(names, o, ref):
        if not torch.equal(a, b):
            bad[n] += 1
            if bad[n] <= 2:
                d = (a != b); idx = torch.nonzero(d)
                print(tag, n, "iter", it, "ndiff", int(d.sum()), "rows", torch.unique(idx[:, 0]).numel(), "first", idx[0].tolist(), flush=True)
print(tag, mode, os.environ.get("CLIFT_X6_TILED"), f"{time.time()-t0:.1f}s mismatches of {iters}:", bad, flush=True)
