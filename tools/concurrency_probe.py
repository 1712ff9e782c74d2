import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine, _lib
dev = "cuda"
M, N, K = 265000, 256, 256
A = [torch.randn(M, K, device=dev) for _ in range(2)]
B = [torch.randn(N, K, device=dev) for _ in range(2)]
Cc = [torch.empty(M, N, device=dev) for _ in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(conc, reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if conc:
            fork = torch.cuda.Event(); fork.record()
            for i in range(2):
                s[i].wait_event(fork); _lib.set_launch_stream(s[i])
                engine.gemm(M, N, K, A[i], K, B[i], K, Cc[i], N)
                _lib.set_launch_stream(None)
                ev = torch.cuda.Event(); ev.record(s[i]); torch.cuda.current_stream().wait_event(ev)
        else:
            for i in range(2):
                engine.gemm(M, N, K, A[i], K, B[i], K, Cc[i], N)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(2):
    print("sequential 2 GEMMs: %.1f us   concurrent 2 GEMMs: %.1f us" % (run(False) * 1e3, run(True) * 1e3))
