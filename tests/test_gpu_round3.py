"""Round-3 GPU parity tests (VERDICT r2 "thin spots"):

  * the Messy-Rooms class count C = 2 takes its own fp32 code path for the semantic head (the output layer is applied inside the last
    hidden layer's kernel, ``clift_xyz_head_last2_fwd``, because C <= 4): forward AND every parameter gradient against the CPU oracle at a
    mid-size anisotropic scene and once at the bench shape (4096 rays x 440 samples, 128^3);
  * the full-size gradient comparison repeated against the oracle run in FLOAT64: how many entries of each gradient fall outside the band
    for (a) the HIP path and (b) the fp32 oracle itself.  fp32 evaluation flips a handful of samples across the 1e-4 activity threshold or a
    ReLU kink whatever the implementation; the test shows that the HIP path's outliers are that effect (same order as the fp32 oracle's own)
    and not a systematic error of the scatter kernels, and holds both to a band tightened to what fp32 itself needs;
  * the persistent fp32x6 layer kernels against fp64 next to the exact kernels on ragged sizes, and a full forward + backward in fp32x6 mode.
"""
import os

import numpy as np
import pytest
import torch

from conftest import grad_close, rel_close
from test_gpu_parity import DEV, _import, _run_forward_backward, build_model, scene

pytestmark = pytest.mark.gpu


def _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, mode, white, dtype=torch.float32, dist_w=3.0):
    Pg = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in P.items()}
    cfg = orender.RenderCfg(aabb.to(dtype), res, density_shift=-3.0, semantic_weight_mode=mode)
    o = orender.render_forward(Pg, rays.to(dtype), cfg, jitter.to(dtype), white)
    L = sum((o[i] * cots[i].to(dtype)).sum() for i in range(3)) + dist_w * o[5]
    L.backward()
    return o, {k: (torch.zeros_like(v) if v.grad is None else v.grad) for k, v in Pg.items()}


# ============================================================================ C = 2: semantic head through the fused output layer
@pytest.mark.parametrize("shape", ["mid", "full"])
@pytest.mark.parametrize("mode", ["softmax", "none"])
def test_forward_backward_vs_oracle_two_classes(shape, mode):
    """C = 2 (dataset/many_object_scenes.py:135-141), E = 3.  mid: grid 40 x 48 x 56, 900 rays; full: 128^3, 4096 rays, S = 440 (one chunk,
    ~250 k active samples).  Outputs to 1e-3 relative (north_star); gradients in the band of test_full_size_backward_vs_oracle."""
    if shape == "full" and mode == "none":
        pytest.skip("the full-size case is run once (softmax, the shipped configuration)")
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    C_, E = 2, 3
    if shape == "mid":
        res, N, aabb, kw = (40, 48, 56), 900, torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]]), dict(amp=2.2, sg=0.4)
    else:
        res, N, aabb, kw = (128, 128, 128), 4096, torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]), dict(img=64, amp=3.0, sg=0.35)
    P, rays, rng = scene(op, orays, 57, res, C_, E, N, **kw)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    o, gref = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, mode, False)
    m = build_model(cl, P, res, C_, E, -3.0, mode)
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode=mode).to(DEV)
    outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    if shape == "full":
        assert int(r.n_samples) == 440
        with torch.no_grad():
            _, ctx = engine.render_forward(m, r, rays.to(DEV), jitter.to(DEV), False)
        assert ctx.M >= 160000, ctx.M
    for a, b, nm in zip(outs[:4], o[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=f"C=2 {nm}")
    rel_close(outs[5], o[5].detach(), 1e-3, what="C=2 dist_reg")
    n = 0
    for k, gr in grads.items():
        got = torch.zeros_like(gref[k]) if gr is None else gr.detach().cpu()
        grad_close(got, gref[k], what=f"C=2 {shape} grad {k}", rtol=2e-3, scale_atol=1e-4,
                   outlier_frac=(5e-3 if k.split(".")[0].endswith(("_plane", "_line")) else 1e-2 if k.startswith("appearance_basis") else 1e-3),
                   outlier_cap=1e-2 if shape == "full" else 1e-3)
        n += 1
    assert n >= 38
    sem_keys = [k for k in grads if k.startswith("render_semantic_mlp")]
    assert sem_keys and all(float(gref[k].abs().max()) > 0 for k in sem_keys)       # the C = 2 head really carries gradient here


# ============================================================================ full size against the oracle in float64
def _outliers(a, b, rtol=2e-3, scale_atol=1e-4):
    a, b = a.double(), b.double()
    mx = float(b.abs().max())
    err = (a - b).abs()
    return int((err > rtol * b.abs() + scale_atol * mx + 1e-12).sum()), float(err.max()) / max(mx, 1e-300)


def test_full_size_gradients_hip_and_fp32_oracle_against_fp64_oracle():
    """4096 rays x 440 samples, 128^3, C = 22: every parameter gradient of (a) the HIP path and (b) the fp32 CPU oracle against the SAME
    oracle evaluated in float64.  Printed side by side (pytest -s; profiles/r03_fp64_outliers.txt); asserted: per tensor the HIP path has no more
    out-of-band entries than 3x the fp32 oracle's + 0.02 % of the tensor (or one row of a 256 x 256 matrix), in total no more than 2x, and its worst entry stays within 3x the fp32 oracle's worst (+2e-3 of the scale) --
    i.e. the outlier allowances of the fp32-vs-fp32 tests are round-off flips that fp32 itself produces, not a property of the kernels."""
    cl, op, orender, ofld, olosses, orays = _import()
    res, C_, E, N = (128, 128, 128), 22, 3, 4096
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P, rays, rng = scene(op, orays, 41, res, C_, E, N, img=64, amp=3.0, sg=0.35)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    o64, g64 = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, "softmax", False, dtype=torch.float64)
    o32, g32 = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, "softmax", False, dtype=torch.float32)
    m = build_model(cl, P, res, C_, E, -3.0, "softmax")
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    for a, b, nm in zip(outs[:4], o64[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=f"{nm} vs fp64 oracle")
    rows, tot_hip, tot_o32, tot_n = [], 0, 0, 0
    for k, gr in grads.items():
        ref = g64[k]
        hip = torch.zeros_like(ref) if gr is None else gr.detach().cpu()
        n_hip, w_hip = _outliers(hip, ref)
        n_o32, w_o32 = _outliers(g32[k], ref)
        rows.append((k, ref.numel(), n_hip, n_o32, w_hip, w_o32))
        tot_hip += n_hip; tot_o32 += n_o32; tot_n += ref.numel()
    print("\n%-46s %9s %8s %8s %10s %10s" % ("gradient (vs fp64 oracle)", "entries", "HIP out", "fp32 out", "HIP worst", "fp32 worst"))
    for k, n, a, b, wa, wb in rows:
        print("%-46s %9d %8d %8d %10.2e %10.2e" % (k, n, a, b, wa, wb))
    print("%-46s %9d %8d %8d" % ("total", tot_n, tot_hip, tot_o32))
    for k, n, n_hip, n_o32, w_hip, w_o32 in rows:
        # (one hidden unit on the other side of its ReLU kink for one sample moves one ROW of the next weight gradient: up to 256 entries)
        assert n_hip <= 3 * n_o32 + max(int(2e-4 * n), 256 if k.endswith(".weight") and n >= 65536 else 1), \
            f"{k}: HIP {n_hip} vs fp32 oracle {n_o32} entries outside the band (of {n})"
        assert w_hip <= 3 * w_o32 + 2e-3, f"{k}: worst HIP error {w_hip:.2e} of the scale vs fp32 oracle {w_o32:.2e}"
        # the allowances of the fp32-vs-fp32 tests (0.5 % tables, 1 % basis matrix, 0.1 % networks), tightened to what fp32 itself needs
        # against the fp64 reference
        head = k.split(".")[0]
        frac = TIGHT_FRAC["plane"] if head.endswith("_plane") else TIGHT_FRAC["line"] if head.endswith("_line") else \
            TIGHT_FRAC["basis"] if k.startswith("appearance_basis") else TIGHT_FRAC["net"]
        assert n_hip <= max(1, int(frac * n)), f"{k}: {n_hip}/{n} outside the tightened band"
    assert tot_hip <= 2 * tot_o32 + int(1e-4 * tot_n)


# measured (profiles/r03_fp64_outliers.txt): planes <= 0.009 % (fp32 oracle 0.013 %), lines <= 0.26 % (0.39 %), basis 0.7 % (0.5 %), networks one
# row of one matrix (43 of 65536); the fp32-vs-fp32 tests allow 0.5 % / 0.5 % / 1 % / 0.1 %
TIGHT_FRAC = {"plane": 5e-4, "line": 5e-3, "basis": 1e-2, "net": 1e-3}


# ============================================================================ fp32x6 persistent layer kernels
@pytest.mark.parametrize("M", [1, 31, 33, 300, 4097, 66001])
def test_fp32x6_persistent_layers_against_fp64(M):
    """csrc/layer_x6.hip (clift_gemm precision 2, N = K = 256): forward (bias + ReLU, padded output pitch untouched) and masked dgrad on rows of
    very different scale (x e^{1.5 N(0,1)}): row-max relative error against fp64 <= 2e-6 and <= 4x the exact-fp32 kernel's + 2e-7; rows are
    independent of how many share the launch (bit-identical between M rows alone and the first M rows of a larger launch)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(3 + M)
    A = (torch.relu(torch.randn(M + 77, 256, generator=g)) * torch.exp(1.5 * torch.randn(M + 77, 1, generator=g))).to(DEV)
    W = (torch.randn(256, 256, generator=g) / 16).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    mk = torch.randn(M + 77, 256, generator=g).to(DEV)
    ref_f = torch.relu(A[:M].double() @ W.double().T + b.double())
    ref_d = (A[:M].double() @ W.double()) * (mk[:M] > 0)
    out = {}
    for mode in ("fp32", "fp32x6"):
        prev = engine.set_mlp_precision(mode)
        try:
            with engine._Precision(engine._PRECISIONS[mode]):
                C1 = torch.full((M, 260), -7.0, device=DEV)
                C2 = torch.full((M, 256), -7.0, device=DEV)
                engine.gemm(M, 256, 256, A, 256, W, 256, C1, 260, bias=b, act=1)
                engine.gemm(M, 256, 256, A, 256, W, 256, C2, 256, b_trans=1, mask=mk, ldmask=256)
                C3 = torch.empty((M + 77, 256), device=DEV)
                engine.gemm(M + 77, 256, 256, A, 256, W, 256, C3, 256, bias=b, act=1)
        finally:
            engine.set_mlp_precision(prev)
        out[mode] = (C1, C2, C3)
    sf = ref_f.abs().amax(1, keepdim=True).clamp_min(1e-30)
    sd = ref_d.abs().amax(1, keepdim=True).clamp_min(1e-30)
    e = {k: (float(((v[0][:, :256].double() - ref_f).abs() / sf).max()), float(((v[1].double() - ref_d).abs() / sd).max())) for k, v in out.items()}
    assert bool((out["fp32x6"][0][:, 256:] == -7.0).all())
    for i in range(2):
        assert e["fp32x6"][i] <= 2e-6 and e["fp32x6"][i] <= 4 * e["fp32"][i] + 2e-7, e
    assert torch.equal(out["fp32x6"][2][:M], out["fp32x6"][0][:, :256])             # a row's bits do not depend on the launch it is in


def test_fp32x6_mode_full_forward_backward_vs_oracle():
    """mlp_dtype fp32x6 through the renderer: outputs 1e-3 relative and every gradient in the band of the exact path's test, C = 22 mid-size."""
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    res, C_, E, N = (40, 48, 56), 22, 3, 900
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P, rays, rng = scene(op, orays, 23, res, C_, E, N, amp=2.2, sg=0.4)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    o, gref = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, "softmax", False, dtype=torch.float64)
    m = build_model(cl, P, res, C_, E, -3.0, "softmax")
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    prev = engine.set_mlp_precision("fp32x6")
    try:
        outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    finally:
        engine.set_mlp_precision(prev)
    for a, b, nm in zip(outs[:4], o[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=f"fp32x6 {nm}")
    for k, gr in grads.items():
        got = torch.zeros_like(gref[k]) if gr is None else gr.detach().cpu()
        grad_close(got, gref[k], what=f"fp32x6 grad {k}", rtol=2e-3, scale_atol=1e-4,
                   outlier_frac=(5e-3 if k.split(".")[0].endswith(("_plane", "_line")) else 1e-2 if k.startswith("appearance_basis") else 1e-3),
                   outlier_cap=1e-3)
