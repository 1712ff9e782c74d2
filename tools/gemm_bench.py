#!/usr/bin/env python3
"""Micro-benchmark of clift_gemm on the shapes of the bench workload (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastive_lift_amd import engine

M = int(sys.argv[1]) if len(sys.argv) > 1 else 265000
stored = len(sys.argv) > 2 and sys.argv[2] == "bf16s"       # bf16 mode with bf16-STORED streamed operands / outputs
if len(sys.argv) > 2:
    engine.set_mlp_precision("bf16" if stored else sys.argv[2])      # fp32 | bf16 | fp32x6 | bf16s
dev = "cuda"
shapes = [("fwd 256x256", M, 256, 256, 0, 0), ("dgrad 256x256", M, 256, 256, 0, 1), ("wgrad 256x256", 256, 256, M, 1, 1),
          ("fwd 152->128", M, 128, 152, 0, 0), ("fwd 128->128", M, 128, 128, 0, 0), ("wgrad 128x128", 128, 128, M, 1, 1), ("wgrad 128x152", 128, 152, M, 1, 1), ("fwd 256->22", M, 22, 256, 0, 0),
          ("fwd 144->27", M, 27, 144, 0, 0), ("dgrad 22->256", M, 256, 22, 0, 1), ("wgrad 22x256", 22, 256, M, 1, 1)]
for name, m, n, k, at, bt in shapes:
    if at:   # wgrad: A = dY (K x m) stored (K, m) row-major; B = X (K x n)
        A = torch.randn(k, (m + 3) // 4 * 4, device=dev); B = torch.randn(k, n, device=dev)
        lda, ldb = A.shape[1], n
    else:
        A = torch.randn(m, (k + 3) // 4 * 4, device=dev)
        B = torch.randn(k, n, device=dev) if bt else torch.randn(n, (k + 3) // 4 * 4, device=dev)
        lda, ldb = A.shape[1], B.shape[1]
    Cm = torch.zeros(m, n, device=dev)
    if stored:       # streamed operands (activations / gradients) and non-accumulated outputs as bf16; weights stay fp32
        if at:
            A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
        else:
            A = A.to(torch.bfloat16)
            Cm = Cm.to(torch.bfloat16) if n % 8 == 0 else Cm
        if (A.shape[1] % 8) or (at and B.shape[1] % 8):
            continue
    kw = dict(a_trans=at, b_trans=bt)
    if not at and bt and n % 8 == 0:      # dgrad of a hidden layer: ReLU mask of the layer input (same storage as the output)
        kw.update(mask=torch.relu(torch.randn(m, n, device=dev)).to(Cm.dtype), ldmask=n)
    if not at and not bt:
        kw.update(bias=torch.randn(n, device=dev), act=1)
    if at:
        kw.update(accumulate=1, split_k=engine._splits(m, n, k))
    for _ in range(3):
        engine.gemm(m, n, k, A, lda, B, ldb, Cm, n, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    reps = 10
    for _ in range(reps):
        engine.gemm(m, n, k, A, lda, B, ldb, Cm, n, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:16s} M={m:7d} N={n:4d} K={k:7d}  {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.1f} TFLOP/s")

# narrow weight gradients go through engine.wgrad (clift_wgrad_narrow: VALU streaming kernel), not clift_gemm
for no, ni in ((22, 256), (3, 256), (3, 128), (27, 144)):
    ldd = (no + 3) // 4 * 4
    dY = torch.randn(M, ldd, device=dev); X = torch.randn(M, ni, device=dev)
    if stored and ni % 8 == 0:
        X = X.to(torch.bfloat16)
    gW = torch.zeros(no, ni, device=dev); gb = torch.zeros(no, device=dev)
    for _ in range(3):
        engine.wgrad(no, ni, M, dY, ldd, X, ni, gW, gb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        engine.wgrad(no, ni, M, dY, ldd, X, ni, gW, gb)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"narrow wgrad {no}x{ni:<4d} M={M:7d}  {ms*1e3:8.1f} us  {X.element_size()*M*ni/ms/1e6:7.1f} GB/s of X")
