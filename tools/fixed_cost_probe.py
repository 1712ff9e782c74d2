"""Fixed cost of the persistent 256 x 256 layer launches: time against the number of 32-row tiles per block (M = 8192 t rows = t tiles in each
of the 256 blocks), so that  time(t) = fixed + t x tile.  MI355X only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine

dev = torch.device("cuda:0")


def timed(fn, n=40):
    """median of per-launch event brackets (a one-off stall -- tens of ms, seen now and then on a fresh box -- would swamp a mean)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[n // 2] * 1e3


g = torch.Generator().manual_seed(0)
W = (torch.randn(256, 256, generator=g) / 16).to(dev)
b = torch.zeros(256, device=dev)
W0, b0 = torch.randn(256, 3, generator=g).to(dev), torch.zeros(256, device=dev)
Wo, bo = (torch.randn(3, 256, generator=g) / 16).to(dev), torch.zeros(3, device=dev)
rows = []
for t in (1, 2, 4, 8, 16):
    M = 8192 * t
    A = torch.relu(torch.randn(M, 256, generator=g)).to(dev)
    d = torch.randn(M, 256, generator=g).to(dev)
    x = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).contiguous().to(dev)
    out, o4 = torch.empty(M, 256, device=dev), torch.empty(M, 4, device=dev)
    gW, gb = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
    gW0, gb0 = torch.zeros(256, 3, device=dev), torch.zeros(256, device=dev)
    r = {
        "fwd": timed(lambda: engine.gemm(M, 256, 256, A, 256, W, 256, out, 256, bias=b, act=1)),
        "gen": timed(lambda: engine.first2(M, x, W0, b0, W, b, None, out)),
        "outv": timed(lambda: engine.last2(M, A, W, b, Wo, bo, out, o4, 4, 0)),
        "dgrad": timed(lambda: engine.gemm(M, 256, 256, d, 256, W, 256, out, 256, b_trans=1, mask=A, ldmask=256)),
        "k3w": timed(lambda: engine.first2_bwd(M, d, W, W0, b0, x, gW0, gb0)),
        "wgrad": timed(lambda: engine.wgrad(256, 256, M, d, 256, A, 256, gW, gb)),
        "empty": timed(lambda: engine.reset_rows_limit(dev)),
    }
    rows.append((t, r))
    print(f"tiles/block {t:3d} (M = {M:6d}): " + "  ".join(f"{k} {v:6.1f}" for k, v in r.items()), flush=True)
for k in rows[0][1]:
    if k == "empty":
        continue
    t1, t2, t16 = rows[0][1][k], rows[1][1][k], rows[4][1][k]
    tile = (t16 - t2) / 14.0
    print(f"{k:6s}: per tile {tile:5.2f} us, fixed = time(1 tile) - tile = {t1 - tile:5.1f} us")
