"""Times of the fused first-two-layers backward (clift_xyz_head_first2_bwd) against the pair it replaces (masked dgrad + K = 3 weight
gradient), and of the output-fused forward kernel (clift_xyz_head_last2_fwd), at the bench's two launch sizes.  MI355X only."""
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
    d = (torch.randn(M, 256, generator=g) * (torch.rand(M, 256, generator=g) > 0.5)).to(dev)
    W = (torch.randn(256, 256, generator=g) / 16).to(dev)
    h = torch.relu(torch.randn(M, 256, generator=g)).to(dev)
    x = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).contiguous().to(dev)
    gW, gb = torch.zeros(256, 3, device=dev), torch.zeros(256, device=dev)
    W0, b0 = torch.randn(256, 3, generator=g).to(dev), (0.5 * torch.randn(256, generator=g)).to(dev)
    dn = torch.empty(M, 256, device=dev)
    t_f = timed(lambda: engine.first2_bwd(M, d, W, W0, b0, x, gW, gb))
    t_d = timed(lambda: engine.gemm(M, 256, 256, d, 256, W, 256, dn, 256, b_trans=1, mask=h, ldmask=256))
    t_k = timed(lambda: call("clift_linear_k3_bwd", ptr(x), ptr(dn), 256, M, 256, ptr(gW), 3, ptr(gb), 0, stream()))
    print(f"M={M}: fused first2_bwd {t_f:7.1f} us   masked dgrad {t_d:7.1f} us + k3 weight gradient {t_k:6.1f} us = {t_d + t_k:7.1f} us")
    gW1, gb1 = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
    t_wg = timed(lambda: engine.first2_wgrad(M, d, W0, b0, x, gW1, gb1))
    t_ws = timed(lambda: engine.wgrad(256, 256, M, d, 256, h, 256, gW1, gb1))
    h2 = torch.empty(M, 256, device=dev)
    t_fk = timed(lambda: engine.first2(M, x, W0, b0, W, gb, h, h2))
    t_fd = timed(lambda: engine.first2(M, x, W0, b0, W, gb, None, h2))
    print(f"M={M}: second-layer weight gradient, activation generated {t_wg:7.1f} us, streamed {t_ws:7.1f} us;  first2 forward h1 kept {t_fk:7.1f} us, not written {t_fd:7.1f} us")
    Wo, bo = (torch.randn(3, 256, generator=g) / 16).to(dev), torch.zeros(3, device=dev)
    b = torch.zeros(256, device=dev)
    hid, out = torch.empty(M, 256, device=dev), torch.empty(M, 4, device=dev)
    t_keep = timed(lambda: engine.last2(M, h, W, b, Wo, bo, hid, out, 4, 0))
    t_drop = timed(lambda: engine.last2(M, h, W, b, Wo, bo, None, out, 4, 0))
    t_plain = timed(lambda: engine.gemm(M, 256, 256, h, 256, W, 256, hid, 256, bias=b, act=1))
    print(f"M={M}: last2 forward hidden kept {t_keep:7.1f} us, dropped {t_drop:7.1f} us; plain forward layer {t_plain:7.1f} us")
