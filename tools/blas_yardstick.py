#!/usr/bin/env python3
"""Yardstick only (never used by the product): what the vendor BLAS behind torch.matmul reaches on the bench's GEMM shapes,
fp32 in / fp32 out, to judge how far k_gemm is from a tuned library kernel on the same chip."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
M = 265000
for name, a_shape, b_shape, tr in [("fwd 256x256", (M, 256), (256, 256), "nt"), ("dgrad 256x256", (M, 256), (256, 256), "nn"),
                                   ("wgrad 256x256", (M, 256), (M, 256), "tn"), ("fwd 128x128", (M, 128), (128, 128), "nt")]:
    A = torch.randn(a_shape, device="cuda"); B = torch.randn(b_shape, device="cuda")
    f = {"nt": lambda: A @ B.T, "nn": lambda: A @ B, "tn": lambda: A.T @ B}[tr]
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * M * a_shape[1] * (b_shape[0] if tr != "tn" else b_shape[1])
    print(f"{name:16s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s   (vendor BLAS via torch.matmul, fp32)")
