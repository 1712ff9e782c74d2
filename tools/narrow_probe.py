"""Times of the narrow-layer stream kernels at the bench's launch sizes (output-layer backward in one pass, masked narrow dgrad,
narrow forward GEMM, narrow weight gradient) with the bytes they move.  MI355X only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine
from contrastive_lift_amd._lib import call, ptr, stream

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (249000, 62000):
    g = torch.Generator().manual_seed(M)
    h = torch.relu(torch.randn(M, 256, generator=g)).to(dev)
    dn = torch.empty(M, 256, device=dev)
    for no, ldd in ((22, 24), (3, 4)):
        d = torch.zeros(M, ldd); d[:, :no] = torch.randn(M, no, generator=g); d = d.to(dev)
        W = (torch.randn(no, 256, generator=g) / 16).to(dev)
        gW, gb = torch.zeros(no, 256, device=dev), torch.zeros(no, device=dev)
        out = torch.empty(M, ldd, device=dev)
        b = torch.zeros(no, device=dev)
        os.environ["CLIFT_NARROW_PREFETCH"] = "0"
        t_bwd0 = timed(lambda: call("clift_out_layer_bwd", ptr(d), ldd, no, ptr(W), 256, ptr(h), 256, M, ptr(dn), 256, ptr(gW), 256, ptr(gb), stream()))
        del os.environ["CLIFT_NARROW_PREFETCH"]
        t_bwd = timed(lambda: call("clift_out_layer_bwd", ptr(d), ldd, no, ptr(W), 256, ptr(h), 256, M, ptr(dn), 256, ptr(gW), 256, ptr(gb), stream()))
        t_dg = timed(lambda: engine.gemm(M, 256, no, d, ldd, W, 256, dn, 256, b_trans=1, mask=h, ldmask=256))
        t_fw = timed(lambda: engine.gemm(M, no, 256, h, 256, W, 256, out, ldd, bias=b))
        t_wg = timed(lambda: engine.wgrad(no, 256, M, d, ldd, h, 256, gW, gb))
        mb = M * 2048 / 1e6
        print(f"M={M} no={no}: out_layer_bwd {t_bwd:6.1f} us ({mb / t_bwd:5.2f} TB/s; same-tile mask fetch {t_bwd0:6.1f} us)  masked dgrad {t_dg:6.1f} us ({mb / t_dg:5.2f} TB/s)  "
              f"forward {t_fw:6.1f} us ({M * 1024 / 1e6 / t_fw:5.2f} TB/s)  weight gradient {t_wg:6.1f} us ({M * 1024 / 1e6 / t_wg:5.2f} TB/s)")
