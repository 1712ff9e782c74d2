"""Per-shape timing of the narrow weight-gradient stream (clift_wgrad_narrow) at the bench's row count."""
import sys, torch
sys.path.insert(0, ".")
from contrastive_lift_amd import engine
M = int(sys.argv[1]) if len(sys.argv) > 1 else 265000
dev = "cuda"
for no, ldd, ni in ((22, 24, 256), (3, 4, 256), (3, 4, 128), (27, 28, 144), (27, 28, 128), (27, 28, 160)):
    dY = torch.zeros((M, ldd), device=dev); dY[:, :no] = torch.randn((M, no), device=dev)
    X = torch.relu(torch.randn((M, ni), device=dev))
    gW = torch.zeros((no, ni), device=dev); gb = torch.zeros((no,), device=dev)
    for _ in range(3):
        engine.wgrad(no, ni, M, dY, ldd, X, ni, gW, gb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        engine.wgrad(no, ni, M, dY, ldd, X, ni, gW, gb)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byt = M * (ni + ldd) * 4
    print(f"no={no} ldd={ldd} ni={ni}: {us:.1f} us, {byt / us / 1e6:.2f} TB/s")
