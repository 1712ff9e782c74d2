"""The exposed neighbour of the GPU-sharing runs (profiles/r03_x6_notes.txt): a process that launches only clift_linear_k3_fwd for `secs` seconds and
compares every result with its first.  Start it, then run the candidate aggressor in ANOTHER process on the same device, e.g.
    python tools/shared_gpu_victim.py 26 &  python bench.py --dtype fp32x6 --no-extras --no-cpu-baseline --steps 1500
(round 3, all three modes' training loops and render loops as aggressors: 0 wrong results of > 100 000 launches each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd._lib import call, ptr, stream
dev = "cuda"; M = 100000; secs = float(sys.argv[1])
g = torch.Generator().manual_seed(1)
b = torch.randn(256, generator=g).to(dev)
xa = torch.zeros(M, 4); xa[:, :3] = torch.rand(M, 3, generator=g) * 2 - 1; xa = xa.to(dev)
W0 = torch.zeros(256, 4); W0[:, :3] = torch.randn(256, 3, generator=g); W0 = W0.to(dev)
h = torch.empty(M, 256, device=dev)
def k3(): call("clift_linear_k3_fwd", ptr(xa), ptr(W0), 4, ptr(b), M, 256, 1, ptr(h), 256, 0, stream())
k3(); ref = h.clone(); torch.cuda.synchronize(); bad = n = 0; t0 = time.time()
while time.time() - t0 < secs:
    k3(); n += 1
    if not torch.equal(h, ref): bad += 1
print(f"victim k3: {bad} wrong of {n}", flush=True)
