#!/usr/bin/env python3
"""Per-launch times of the appearance head's front end (forward: gather / basis / encode vs clift_app_front_fwd; backward: encode_bwd / basis
weight gradient / dF GEMM / scatter) on the bench scene (128^3, 4096 rays).
   python tools/app_probe.py [path of an alternative libclift.so]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrastive_lift_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from contrastive_lift_amd import engine, synthetic
from contrastive_lift_amd._lib import call, ptr, stream

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device=dev)
b = synthetic.make_batches(pool, 4096, 1024, 22, 25, seed=100, device=dev)
rays = b[0]["rays"]
jit = torch.rand(rays.shape[0], device=dev)
out, ctx = engine.render_forward(model, renderer, rays, jit, False, grad_heads=("app", "sem"))
torch.cuda.synchronize()
M = ctx.M
views = model.named_views()
va = engine.vm_struct(views, "appearance", ctx.res)
Wb = views["appearance_basis_mat.weight"]
nf, nc = Wb.shape
ldb = engine._pitch(Wb)
ldx = ctx.ldx


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


F = torch.empty(M, nc, device=dev); xa = torch.empty(M, 4, device=dev); feat = torch.empty(M, 28, device=dev); X = torch.empty(M, ldx, device=dev)
S = int(renderer.n_samples)
t_g = timeit(lambda: call("clift_app_gather_fwd", C.byref(ctx.ms), C.byref(va), ptr(rays), ptr(jit), ptr(ctx.act_idx), M, ptr(F), ptr(xa), stream()))
with engine.exact_fp32():
    t_b = timeit(lambda: engine.gemm(M, nf, nc, F, nc, Wb, ldb, feat, 28))
t_e = timeit(lambda: call("clift_app_encode_fwd", ptr(feat), 28, nf, model.pe_feat, model.pe_view, ptr(rays), ptr(ctx.act_idx), S, M, ptr(X), ldx, 0, stream()))
t_f = timeit(lambda: call("clift_app_front_fwd", C.byref(ctx.ms), C.byref(va), ptr(rays), ptr(jit), ptr(ctx.act_idx), M, ptr(Wb), ldb, nf,
                          model.pe_feat, model.pe_view, ptr(xa), ptr(feat), 28, ptr(X), ldx, ptr(F), stream()))
t_f0 = timeit(lambda: call("clift_app_front_fwd", C.byref(ctx.ms), C.byref(va), ptr(rays), ptr(jit), ptr(ctx.act_idx), M, ptr(Wb), ldb, nf,
                           model.pe_feat, model.pe_view, ptr(xa), ptr(feat), 28, ptr(X), ldx, None, stream()))
print(f"M {M}  forward: gather {t_g:6.1f} + basis {t_b:6.1f} + encode {t_e:6.1f} = {t_g + t_b + t_e:6.1f} us   fused (F written) {t_f:6.1f}   fused (no F) {t_f0:6.1f} us", flush=True)

# backward
dX = torch.randn(M, ldx, device=dev); dfeat = torch.empty(M, 28, device=dev); dF = torch.empty(M, nc, device=dev)
gWb = torch.zeros(nf, ldb, device=dev)
gv = model.named_grad_views()
model.xcd_workspace_for("appearance")
ga = engine.vm_grad_struct(model, gv, "appearance")
t_eb = timeit(lambda: call("clift_app_encode_bwd", ptr(feat), 28, nf, model.pe_feat, ptr(dX), ldx, M, ptr(dfeat), 28, stream()))
t_wb = timeit(lambda: call("clift_wgrad_narrow", ptr(dfeat), 28, nf, ptr(F), nc, nc, M, ptr(gWb), ldb, None, 0, stream()))
with engine.exact_fp32():
    t_df = timeit(lambda: engine.gemm(M, nc, nf, dfeat, 28, Wb, ldb, dF, nc, b_trans=1))
t_sc = timeit(lambda: call("clift_app_gather_bwd", C.byref(ctx.ms), C.byref(va), C.byref(ga), ptr(rays), ptr(jit), ptr(ctx.act_idx), M, ptr(dF), ptr(xa), stream()))
line = f"backward: encode_bwd {t_eb:6.1f}  basis wgrad {t_wb:6.1f} + dF gemm {t_df:6.1f} + scatter {t_sc:6.1f} = {t_wb + t_df + t_sc:6.1f} us"
print(line, flush=True)
