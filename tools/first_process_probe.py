"""Where does the one-off stall of the FIRST GPU process on a fresh box land?  Times 400 training steps of the bench workload one by one
(synchronising each) and prints the outliers with their wall-clock offsets.  Run as the first GPU command of a gpurun call."""
import sys, time
t_start = time.time()
import torch
sys.path.insert(0, ".")
from contrastive_lift_amd import synthetic
from contrastive_lift_amd.trainer import HotPathTrainer, default_config
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
t_gpu = time.time()
print(f"first GPU touch at +{t_gpu - t_start:.1f} s after process start", flush=True)
model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device=dev)
cfg = default_config(chunk=0, instance_optimization_epoch=0, late_semantic_optimization=0, mlp_dtype="fp32", nosync=False)
tr = HotPathTrainer(model, renderer, cfg, current_epoch=4)
batches = [synthetic.make_batches(pool, 4096, 1024, 22, 25, seed=100 + i, device=dev) for i in range(4)]
ts = []
for i in range(400):
    torch.cuda.synchronize()
    t0 = time.time()
    tr.training_step(batches[i % 4])
    torch.cuda.synchronize()
    ts.append((t0 - t_gpu, time.time() - t0))
med = sorted(d for _, d in ts)[len(ts) // 2]
print(f"median step {med * 1e3:.2f} ms")
for i, (at, d) in enumerate(ts):
    if d > 1.5 * med:
        print(f"step {i}: {d * 1e3:.1f} ms at +{at:.2f} s after the first GPU touch")
