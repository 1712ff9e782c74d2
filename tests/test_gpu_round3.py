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

from conftest import grad_close, rel_close, usable_cores
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
    torch.set_num_threads(usable_cores())
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
                   outlier_cap=1e-2 if shape == "full" else 2e-3)      # (worst single entry against the fp32 CPU oracle, itself rounded: 1.2e-3 of the scale seen on appearance_plane.1)
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
    """4096 rays x 440 samples, 128^3, C = 22: every parameter gradient of (a) the HIP path (exact fp32, and once more in fp32x6 mode) and (b) the
    fp32 CPU oracle against the SAME oracle evaluated in float64.  Printed side by side (pytest -s; profiles/r03_fp64_outliers.txt); asserted: per tensor the HIP path has no more
    out-of-band entries than 3x the fp32 oracle's + 0.02 % of the tensor (or one row of a 256 x 256 matrix), in total no more than 2x, and its worst entry stays within 3x the fp32 oracle's worst (+2e-3 of the scale) --
    i.e. the outlier allowances of the fp32-vs-fp32 tests are round-off flips that fp32 itself produces, not a property of the kernels."""
    cl, op, orender, ofld, olosses, orays = _import()
    res, C_, E, N = (128, 128, 128), 22, 3, 4096
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P, rays, rng = scene(op, orays, 41, res, C_, E, N, img=64, amp=3.0, sg=0.35)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    torch.set_num_threads(usable_cores())
    o64, g64 = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, "softmax", False, dtype=torch.float64)
    o32, g32 = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, "softmax", False, dtype=torch.float32)
    m = build_model(cl, P, res, C_, E, -3.0, "softmax")
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    from contrastive_lift_amd import engine
    with engine.exact_fp32():
        outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    for a, b, nm in zip(outs[:4], o64[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=f"{nm} vs fp64 oracle")
    # the same step in fp32x6 mode (the 256 x 256 forward / dgrad layers as six bf16 products, csrc/layer_x6.hip): a third column
    prev = engine.set_mlp_precision("fp32x6")
    try:
        m6 = build_model(cl, P, res, C_, E, -3.0, "softmax")
        outs6, grads6 = _run_forward_backward(cl, m6, r, rays, jitter, False, cots + [3.0])
    finally:
        engine.set_mlp_precision(prev)
    for a, b, nm in zip(outs6[:4], o64[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=f"fp32x6 {nm} vs fp64 oracle")
    rows, tot_hip, tot_o32, tot_x6, tot_n = [], 0, 0, 0, 0
    for k, gr in grads.items():
        ref = g64[k]
        hip = torch.zeros_like(ref) if gr is None else gr.detach().cpu()
        x6 = torch.zeros_like(ref) if grads6[k] is None else grads6[k].detach().cpu()
        n_hip, w_hip = _outliers(hip, ref)
        n_o32, w_o32 = _outliers(g32[k], ref)
        n_x6, w_x6 = _outliers(x6, ref)
        rows.append((k, ref.numel(), n_hip, n_o32, w_hip, w_o32))
        assert n_x6 <= 3 * n_o32 + max(int(2e-4 * ref.numel()), 256 if k.endswith(".weight") and ref.numel() >= 65536 else 1), \
            f"{k}: fp32x6 {n_x6} vs fp32 oracle {n_o32} entries outside the band (of {ref.numel()})"
        assert w_x6 <= 3 * w_o32 + 2e-3, f"{k}: worst fp32x6 error {w_x6:.2e} of the scale vs fp32 oracle {w_o32:.2e}"
        tot_hip += n_hip; tot_o32 += n_o32; tot_x6 += n_x6; tot_n += ref.numel()
        rows[-1] = rows[-1] + (n_x6, w_x6)
    print("\n%-46s %9s %8s %8s %8s %10s %10s %10s" % ("gradient (vs fp64 oracle)", "entries", "HIP out", "fp32 out", "x6 out", "HIP worst", "fp32 worst", "x6 worst"))
    for k, n, a, b, wa, wb, c, wc in rows:
        print("%-46s %9d %8d %8d %8d %10.2e %10.2e %10.2e" % (k, n, a, b, c, wa, wb, wc))
    print("%-46s %9d %8d %8d %8d" % ("total", tot_n, tot_hip, tot_o32, tot_x6))
    assert tot_x6 <= 2 * tot_o32 + int(1e-4 * tot_n)
    rows = [x[:6] for x in rows]
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


# ============================================================================ first two layers' backward in one launch
def _first2_case(M, seed, cap=None):
    from contrastive_lift_amd._lib import call, ptr, stream
    R = cap or M
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(R, 256, generator=g) * (torch.rand(R, 256, generator=g) > 0.4)
    W1 = torch.randn(256, 256, generator=g) / 16
    W0, b0 = torch.randn(256, 3, generator=g), 0.5 * torch.randn(256, generator=g)
    x4 = torch.cat([torch.rand(R, 3, generator=g) * 2 - 1, torch.zeros(R, 1)], 1).contiguous()
    dev = {k: v.to(DEV) for k, v in dict(d=d, W1=W1, W0=W0, b0=b0, x4=x4).items()}
    h1 = torch.empty(R, 256, device=DEV)          # the forward's first-layer activation (its sign is what the fused kernel has to reproduce)
    call("clift_linear_k3_fwd", ptr(dev["x4"]), ptr(dev["W0"]), 3, ptr(dev["b0"]), R, 256, 1, ptr(h1), 256, 0, stream())
    dH1 = (d[:M].double() @ W1.double()) * (h1[:M].cpu() > 0)
    return dev, h1, dH1.t() @ x4[:M, :3].double(), dH1.sum(0)


@pytest.mark.parametrize("M", [1, 31, 32, 33, 64, 4097, 66001, 249000])
def test_first2_bwd_against_fp64_and_the_unfused_pair(M):
    """clift_xyz_head_first2_bwd (ABI 11): gW0 / gb0 of the K = 3 layer from dH2, W1, (W0, b0) and the positions, without writing
    dH1 = (h1 > 0) . (dH2 W1) and without reading h1, against the same sums in float64 (mask = sign of the activation the forward kernel
    produced) and against the unfused pair (clift_gemm masked dgrad + clift_linear_k3_bwd).  Ragged row counts (single row, one short of /
    one past a 32-row tile, one past a block boundary) and the bench size; the gradients are ACCUMULATED into (a second call doubles them).
    Band: 2e-5 of each tensor's largest entry (fp32 sums over M rows)."""
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    t, h1, gW_ref, gb_ref = _first2_case(M, 1000 + M)
    gW, gb = torch.zeros(256, 3, device=DEV), torch.zeros(256, device=DEV)
    engine.first2_bwd(M, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gW, gb)
    torch.cuda.synchronize()
    for got, ref, nm in ((gW, gW_ref, "gW0"), (gb, gb_ref, "gb0")):
        err = float((got.double().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        assert err < 2e-5, (M, nm, err)
    dn = torch.empty(M, 256, device=DEV)                                              # the unfused pair on the same inputs
    with engine.exact_fp32():
        engine.gemm(M, 256, 256, t["d"], 256, t["W1"], 256, dn, 256, b_trans=1, mask=h1, ldmask=256)
    gW2, gb2 = torch.zeros(256, 3, device=DEV), torch.zeros(256, device=DEV)
    call("clift_linear_k3_bwd", ptr(t["x4"]), ptr(dn), 256, M, 256, ptr(gW2), 3, ptr(gb2), 0, stream())
    for got, ref, nm in ((gW, gW2, "gW0"), (gb, gb2, "gb0")):
        err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        assert err < 2e-5, (M, nm, "fused vs unfused", err)
    engine.first2_bwd(M, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gW, gb)          # accumulation
    err = float((gW.double().cpu() - 2 * gW_ref).abs().max()) / max(float(gW_ref.abs().max()), 1e-30)
    assert err < 4e-5, (M, "accumulate", err)


def test_first2_bwd_under_a_device_side_row_limit():
    """Sync-free step: the launch is sized by a capacity and the kernels (both fused backward kernels) read the true row count from device
    memory -- rows past it (NaN here) must not reach the sums."""
    from contrastive_lift_amd import engine
    cap, M = 9000, 5003
    t, h1, gW_ref, gb_ref = _first2_case(M, 5, cap=cap)
    t["d"][M:] = float("nan"); t["x4"][M:] = float("nan")
    gW, gb = torch.zeros(256, 3, device=DEV), torch.zeros(256, device=DEV)
    lim = engine.rows_limit(gW.device)
    lim[0:1].fill_(M)
    try:
        engine.first2_bwd(cap, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gW, gb)
        torch.cuda.synchronize()
    finally:
        engine.reset_rows_limit(gW.device)
    assert bool(torch.isfinite(gW).all()) and bool(torch.isfinite(gb).all())
    err = float((gW.double().cpu() - gW_ref).abs().max()) / float(gW_ref.abs().max())
    assert err < 2e-5, err
    # the generating weight gradient under the same limit
    gW1, gb1 = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
    lim[0:1].fill_(M)
    try:
        engine.first2_wgrad(cap, t["d"], t["W0"], t["b0"], t["x4"], gW1, gb1)
        torch.cuda.synchronize()
    finally:
        engine.reset_rows_limit(gW.device)
    ref1 = t["d"][:M].double().cpu().t() @ h1[:M].double().cpu()
    assert bool(torch.isfinite(gW1).all()) and bool(torch.isfinite(gb1).all())
    err = float((gW1.double().cpu() - ref1).abs().max()) / float(ref1.abs().max())
    assert err < 2e-5, err


def test_first2_backward_kernels_with_padded_pitches():
    """The two fused kernels through the C ABI with every pitch larger than its row (dH2 260, W1 264, W0 / gW0 4, gW1 272): pad columns are
    neither read into the sums nor written."""
    from contrastive_lift_amd._lib import call, ptr, stream
    M = 7001
    t, h1, gW0_ref, gb0_ref = _first2_case(M, 31)
    dp = torch.full((M, 260), float("nan"), device=DEV); dp[:, :256] = t["d"]
    W1p = torch.full((256, 264), float("nan"), device=DEV); W1p[:, :256] = t["W1"]
    W0p = torch.full((256, 4), float("nan"), device=DEV); W0p[:, :3] = t["W0"]
    gW0 = torch.full((256, 4), -7.0, device=DEV); gW0[:, :3] = 0
    gb0 = torch.zeros(256, device=DEV)
    call("clift_xyz_head_first2_bwd", ptr(dp), 260, ptr(W1p), 264, ptr(W0p), 4, ptr(t["b0"]), ptr(t["x4"]), M, ptr(gW0), 4, ptr(gb0), stream())
    gW1 = torch.full((256, 272), -7.0, device=DEV); gW1[:, :256] = 0
    gb1 = torch.zeros(256, device=DEV)
    call("clift_xyz_head_first2_wgrad", ptr(dp), 260, ptr(W0p), 4, ptr(t["b0"]), ptr(t["x4"]), M, ptr(gW1), 272, ptr(gb1), stream())
    torch.cuda.synchronize()
    assert bool((gW0[:, 3] == -7.0).all()) and bool((gW1[:, 256:] == -7.0).all())
    err = float((gW0[:, :3].double().cpu() - gW0_ref).abs().max()) / float(gW0_ref.abs().max())
    assert err < 2e-5, err
    ref1 = t["d"][:M].double().cpu().t() @ h1[:M].double().cpu()
    err = float((gW1[:, :256].double().cpu() - ref1).abs().max()) / float(ref1.abs().max())
    assert err < 2e-5, err
    assert float((gb1.double().cpu() - t["d"].double().cpu().sum(0)).abs().max()) / float(t["d"].double().cpu().sum(0).abs().max()) < 2e-5


def test_fused_head_entry_points_reject_bad_arguments():
    """Error behaviour of the ABI 11 - 13 entry points: misaligned rows, short pitches and missing outputs come back as a non-zero return with
    a message (CliftError in the binding), nothing is launched."""
    from contrastive_lift_amd import _lib
    from contrastive_lift_amd._lib import call, ptr, stream
    M = 100
    d = torch.zeros(M, 256, device=DEV); W = torch.zeros(256, 256, device=DEV); W0 = torch.zeros(256, 3, device=DEV); b0 = torch.zeros(256, device=DEV)
    x4 = torch.zeros(M + 1, 4, device=DEV); gW0 = torch.zeros(256, 3, device=DEV); gb0 = torch.zeros(256, device=DEV); gW1 = torch.zeros(256, 256, device=DEV)
    h2 = torch.zeros(M, 256, device=DEV)
    with pytest.raises(_lib.CliftError, match="first2_bwd"):            # dH2 pitch below 256
        call("clift_xyz_head_first2_bwd", ptr(d), 128, ptr(W), 256, ptr(W0), 3, ptr(b0), ptr(x4), M, ptr(gW0), 3, ptr(gb0), stream())
    with pytest.raises(_lib.CliftError, match="first2_bwd"):            # no gradient output
        call("clift_xyz_head_first2_bwd", ptr(d), 256, ptr(W), 256, ptr(W0), 3, ptr(b0), ptr(x4), M, None, 3, ptr(gb0), stream())
    with pytest.raises(_lib.CliftError, match="first2_wgrad"):          # positions not 16-byte aligned
        call("clift_xyz_head_first2_wgrad", ptr(d), 256, ptr(W0), 3, ptr(b0), x4.data_ptr() + 4, M, ptr(gW1), 256, None, stream())
    with pytest.raises(_lib.CliftError, match="first2_x6_fwd"):         # output pitch not a multiple of 4
        call("clift_xyz_head_first2_x6_fwd", ptr(x4), ptr(W0), 3, ptr(b0), ptr(W), 256, ptr(b0), M, ptr(h2), 258, None, stream())
    with pytest.raises(_lib.CliftError, match="grad_shards"):
        call("clift_grad_shards_begin", None, None, None, 0, stream())
    assert float(gW0.abs().max()) == 0.0 and float(gW1.abs().max()) == 0.0
    call("clift_xyz_head_first2_bwd", ptr(d), 256, ptr(W), 256, ptr(W0), 3, ptr(b0), ptr(x4), 0, ptr(gW0), 3, ptr(gb0), stream())      # M = 0: a no-op


@pytest.mark.parametrize("M", [1, 63, 64, 65, 4097, 66001, 249000])
def test_first2_wgrad_against_fp64_and_the_streamed_form(M):
    """clift_xyz_head_first2_wgrad (ABI 11): the second layer's weight / bias gradient with its input, relu(W0 x + b0), generated in-kernel,
    against float64 sums over the activation the forward kernel produced and against clift_gemm's weight gradient streaming that stored
    activation.  Ragged row counts around the 64-row tile and the block boundary, and the bench size; accumulating.  Band 2e-5 of the largest entry."""
    from contrastive_lift_amd import engine
    t, h1, _, _ = _first2_case(M, 2000 + M)
    ref = t["d"][:M].double().cpu().t() @ h1[:M].double().cpu()
    gb_ref = t["d"][:M].double().cpu().sum(0)
    gW, gb = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
    engine.first2_wgrad(M, t["d"], t["W0"], t["b0"], t["x4"], gW, gb)
    torch.cuda.synchronize()
    for got, r, nm in ((gW, ref, "gW1"), (gb, gb_ref, "gb1")):
        err = float((got.double().cpu() - r).abs().max()) / max(float(r.abs().max()), 1e-30)
        assert err < 2e-5, (M, nm, err)
    gW2, gb2 = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
    with engine.exact_fp32():
        engine.wgrad(256, 256, M, t["d"], 256, h1, 256, gW2, gb2)
    err = float((gW - gW2).abs().max()) / max(float(gW2.abs().max()), 1e-30)
    assert err < 2e-5, (M, "generated vs streamed", err)
    engine.first2_wgrad(M, t["d"], t["W0"], t["b0"], t["x4"], gW, gb)
    err = float((gW.double().cpu() - 2 * ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    assert err < 4e-5, (M, "accumulate", err)


def test_head_backward_without_the_first_activation_matches_the_stored_form():
    """xyz_mlp_fwd / xyz_mlp_bwd of a 5-layer head (3 -> 256 -> 256 -> 256 -> 256 -> 22) with the fused first-two-layers backward (the forward
    keeps no first-layer activation) against the same head with engine.KEEP_FIRST_ACT set (activation stored, masked dgrad + K = 3 weight
    gradient as separate launches): every parameter gradient within 2e-5 of its scale; the forward outputs bit-identical."""
    from contrastive_lift_amd import engine
    M, C_ = 20011, 22
    g = torch.Generator().manual_seed(77)
    dims = [(256, 3), (256, 256), (256, 256), (256, 256), (C_, 256)]
    layers = [((torch.randn(o, i, generator=g) / (i ** 0.5)).to(DEV), (0.1 * torch.randn(o, generator=g)).to(DEV)) for o, i in dims]
    xa = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).contiguous().to(DEV)
    dpre = torch.zeros(M, 24)
    dpre[:, :C_] = torch.randn(M, C_, generator=g)
    dpre = dpre.to(DEV)
    res = {}
    prev = engine.KEEP_FIRST_ACT
    try:
        for flag in (True, False):
            engine.KEEP_FIRST_ACT = not flag
            out = torch.zeros(M, 24, device=DEV)
            gl = [(torch.zeros_like(W), torch.zeros_like(b)) for W, b in layers]
            with engine.exact_fp32():
                acts = engine.xyz_mlp_fwd(layers, xa, M, out, 24)
                assert (acts[0] is None) == flag
                engine.xyz_mlp_bwd(layers, gl, xa, acts, dpre.clone(), M)
            torch.cuda.synchronize()
            res[flag] = (out, gl)
    finally:
        engine.KEEP_FIRST_ACT = prev
    assert torch.equal(res[True][0], res[False][0])
    for (a, ab), (b, bb) in zip(res[True][1], res[False][1]):
        for x, y in ((a, b), (ab, bb)):
            err = float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
            assert err < 2e-5, (tuple(x.shape), err)


@pytest.mark.parametrize("M", [1, 31, 32, 33, 4097, 66001, 249000])
def test_first2_x6_forward_is_the_unfused_pair_bit_for_bit(M):
    """clift_xyz_head_first2_x6_fwd (ABI 13): the K = 3 layer generated inside the fp32x6 layer kernel.  The generated values are the bits
    clift_linear_k3_fwd writes (same FMA order) and the split / MFMA path is the same, so the result is BIT-IDENTICAL to that launch followed by
    clift_gemm(precision 2) -- and within 2e-6 (row-max relative) of float64."""
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    t, h1, _, _ = _first2_case(M, 3000 + M)
    g = torch.Generator().manual_seed(9)
    b1 = (0.1 * torch.randn(256, generator=g)).to(DEV)
    h2 = torch.full((M, 256), -7.0, device=DEV)
    engine.first2_x6(M, t["x4"], t["W0"], t["b0"], t["W1"], b1, h2)
    ref = torch.empty(M, 256, device=DEV)
    with engine._Precision(2):
        engine.gemm(M, 256, 256, h1, 256, t["W1"], 256, ref, 256, bias=b1, act=1)
    torch.cuda.synchronize()
    assert torch.equal(h2, ref)
    r64 = torch.relu(h1[:M].double() @ t["W1"].double().t() + b1.double())
    err = float(((h2.double() - r64).abs() / r64.abs().amax(1, keepdim=True).clamp_min(1e-30)).max())
    assert err <= 2e-6, err


@pytest.mark.parametrize("M", [4096, 4097, 5000, 66001, 249000])
def test_fp32x6_weight_gradient_against_fp64(M):
    """csrc/layer_x6w.hip (fp32x6 mode): gW += dY^T X, gb += column sums of dY for the 256 x 256 layers as six bf16 products of exactly split
    operands, against float64 next to the exact quadrant kernel: error (of the largest entry) <= 2e-6 and <= 4x the exact kernel's + 2e-7;
    accumulating (a second launch doubles the result)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(40 + M)
    dY = (torch.randn(M, 256, generator=g) * (torch.rand(M, 256, generator=g) > 0.5) * torch.exp(torch.randn(M, 1, generator=g))).to(DEV)
    X = torch.relu(torch.randn(M, 256, generator=g)).to(DEV)
    ref = dY.double().cpu().t() @ X.double().cpu()
    rb = dY.double().cpu().sum(0)
    out = {}
    for mode in ("fp32", "fp32x6"):
        gW, gb = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
        with engine._Precision(engine._PRECISIONS[mode]):
            engine.wgrad(256, 256, M, dY, 256, X, 256, gW, gb)
            if mode == "fp32x6":
                gW2, gb2 = gW.clone(), gb.clone()
                engine.wgrad(256, 256, M, dY, 256, X, 256, gW2, gb2)
        torch.cuda.synchronize()
        out[mode] = (float((gW.double().cpu() - ref).abs().max()) / float(ref.abs().max()), float((gb.double().cpu() - rb).abs().max()) / float(rb.abs().max()))
    assert out["fp32x6"][0] <= 2e-6 and out["fp32x6"][0] <= 4 * out["fp32"][0] + 2e-7, out
    assert out["fp32x6"][1] <= 4e-6, out
    assert float((gW2.double().cpu() - 2 * ref).abs().max()) / float(ref.abs().max()) <= 4e-6


def test_fp32x6_mode_full_forward_backward_vs_oracle():
    """mlp_dtype fp32x6 through the renderer, C = 22 mid-size, against the oracle in FLOAT64, next to the exact-fp32 path on the same inputs:
    outputs 1e-3 relative; per gradient tensor the number of entries outside the band (2e-3 relative + 1e-4 of the scale: these are samples
    whose hidden unit lands on the other side of a ReLU kink, the kernels' own errors are ~1e-6) is of the exact path's order -- at most
    2x its count + 0.1 % of the tensor -- and the worst entry within 1e-3 of the scale."""
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    res, C_, E, N = (40, 48, 56), 22, 3, 900
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P, rays, rng = scene(op, orays, 23, res, C_, E, N, amp=2.2, sg=0.4)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    o, gref = _oracle_run(op, orender, P, rays, jitter, cots, aabb, res, "softmax", False, dtype=torch.float64)
    res_mode = {}
    for mode in ("fp32", "fp32x6"):
        m = build_model(cl, P, res, C_, E, -3.0, "softmax")
        r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
        with engine._Precision(engine._PRECISIONS[mode]):
            res_mode[mode] = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    outs, grads = res_mode["fp32x6"]
    for a, b, nm in zip(outs[:4], o[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=f"fp32x6 {nm}")
    tot = {"fp32": 0, "fp32x6": 0}
    for k, gr in grads.items():
        got = torch.zeros_like(gref[k]) if gr is None else gr.detach().cpu()
        g0 = res_mode["fp32"][1][k]
        exact = torch.zeros_like(gref[k]) if g0 is None else g0.detach().cpu()
        n6, w6 = _outliers(got, gref[k])
        n0, w0 = _outliers(exact, gref[k])
        tot["fp32"] += n0; tot["fp32x6"] += n6
        # one hidden unit on the other side of its kink for ONE sample moves one row of the next layer's weight gradient (up to 256 entries):
        # count the rows the out-of-band entries sit in
        allow = max(1, int(1e-3 * gref[k].numel()))
        if n6 > 2 * n0 + allow and gref[k].dim() == 2 and gref[k].shape[1] >= 128:
            ref = gref[k].double()
            bad = ((got.double() - ref).abs() > 2e-3 * ref.abs() + 1e-4 * float(ref.abs().max()) + 1e-12)
            rows_hit = int(bad.any(1).sum())
            print(f"{k}: {n6} entries outside the band in {rows_hit} row(s) (exact path: {n0})")
            assert rows_hit <= 2, f"{k}: fp32x6 {n6} entries outside the band spread over {rows_hit} rows (exact {n0})"
        else:
            assert n6 <= 2 * n0 + allow, f"{k}: fp32x6 {n6} vs exact {n0} entries outside the band"
        assert w6 <= max(1e-3, 3 * w0), f"{k}: worst fp32x6 error {w6:.2e} of the scale (exact {w0:.2e})"
    print("entries outside the band vs the fp64 oracle, all gradients: exact fp32", tot["fp32"], " fp32x6", tot["fp32x6"])
    assert tot["fp32x6"] <= 2 * tot["fp32"] + 600


# ============================================================================ bf16 mode (configs[2]) against the ORACLE at the Messy-Rooms shape
def test_bf16_mode_against_the_oracle_at_the_messy_rooms_shape():
    _bf16_against_the_oracle((32, 32, 32), 768, 512, 60, 10)


def test_bf16_mode_against_the_oracle_at_the_bench_shape():
    """The same at the bench shape -- 128^3 grid, 4096 + 1024 rays, S = 440 -- over six full training steps (the CPU oracle
    needs ~5 s for one): PSNR within 0.1 dB after every second step."""
    _bf16_against_the_oracle((128, 128, 128), 4096, 1024, 6, 2)


def _bf16_against_the_oracle(res, B, Bi, steps, every):
    """BASELINE configs[2]: Messy-Rooms class count (C = 2: background / foreground, dataset/many_object_scenes.py:135-141), 25 instance
    ids, slow-fast contrastive head, bf16 MLP operands.  `steps` full training steps (main pass + instance pass, both optimizers, EMA) from
    identical weights / batches / jitter through the HIP trainer in bf16 mode and through the fp32 CPU oracle:
      * rendered rgb / semantics / instance features of the first step within 2e-2 of each tensor's scale (bf16 has 8 significant bits:
        the 1e-3 of the fp32 path is not expected, SURVEY 7 'bf16 tolerance'),
      * PSNR on the training rays within 0.1 dB of the oracle's at every checkpoint (north_star), semantic loss within 5 %,
      * the slow-fast loss within 10 % + 5e-3."""
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    from oracle.train_step import CpuTrainer
    C_, E = 2, 3
    torch.set_num_threads(usable_cores())
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P, rays, rng = scene(op, orays, 73, res, C_, E, B + Bi, amp=2.6, sg=0.4, img=(48 if B + Bi <= 3 * 48 * 48 else 96))
    rays_main, rays_inst = rays[:B].contiguous(), rays[B:].contiguous()
    d = rays_main[:, 3:6]
    rgbs = (0.5 + 0.5 * torch.sin(3.0 * d + torch.tensor([0.0, 1.0, 2.0]))).contiguous()
    fg = torch.sigmoid(6.0 * torch.sin(2.0 * d[:, 0]) * torch.cos(3.0 * d[:, 1]))
    probs = torch.stack([1 - fg, fg], -1).contiguous()
    conf = torch.from_numpy(rng.uniform(0.5, 1, B).astype(np.float32))
    # 25 instance ids: five equally populated bands along x, each cut into five equally populated bands along y (rank-based, so every id occurs)
    di = rays_inst[:, 3:6]
    order_x = torch.argsort(di[:, 0])
    band = torch.empty(Bi, dtype=torch.long)
    band[order_x] = torch.arange(Bi) * 5 // Bi
    labels = torch.empty(Bi, dtype=torch.long)
    for bnd in range(5):
        idx = torch.nonzero(band == bnd).reshape(-1)
        oy = idx[torch.argsort(di[idx, 1])]
        labels[oy] = 1 + bnd * 5 + torch.arange(oy.numel()) * 5 // max(1, oy.numel())
    labels = labels.contiguous()
    assert len(torch.unique(labels)) == 25
    iconf = torch.from_numpy(rng.uniform(0.5, 1, Bi).astype(np.float32))
    m = build_model(cl, P, res, C_, E, -3.0)
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    try:
        tr = HotPathTrainer(m, r, default_config(chunk=0, instance_optimization_epoch=0, late_semantic_optimization=0, mlp_dtype="bf16"), current_epoch=4)
        assert engine.MLP_PRECISION == 1
        ct = CpuTrainer(P, orender.RenderCfg(aabb, res, density_shift=-3.0), chunk=4096, epoch=4)
        batch0 = dict(rays=rays_main.to(DEV), rgbs=rgbs.to(DEV), probabilities=probs.to(DEV), confidences=conf.to(DEV), mask=None)
        ibatch = [dict(rays=rays_inst.to(DEV), instances=labels.to(DEV), confidences=iconf.to(DEV))]
        psnr = lambda a, b: float(-10.0 * torch.log10(((a - b) ** 2).mean()))
        # element-wise agreement of one forward (before any update) with the oracle
        jit0 = torch.from_numpy(rng.uniform(0, 1, B).astype(np.float32))
        o = orender.render_forward(op.clone_params(P), rays_main, orender.RenderCfg(aabb, res, density_shift=-3.0), jit0, False)
        with torch.no_grad():
            og, _ = engine.render_forward(m, r, rays_main.to(DEV), jit0.to(DEV), False)
        for nm, a, b in (("rgb", og["rgb"], o[0]), ("semantics", og["semantics"], o[1]), ("instances", og["instances"], o[2])):
            err = float((a.cpu() - b.detach()).abs().max()) / max(1e-6, float(b.abs().max()))
            assert err < 2e-2, (nm, err)
        worst = 0.0
        for step in range(steps):
            jit = torch.from_numpy(rng.uniform(0, 1, B).astype(np.float32))
            jit_i = torch.from_numpy(rng.uniform(0, 1, Bi).astype(np.float32))
            white = bool(step % 3 == 0)
            oc = ct.main_pass(rays_main, rgbs, probs, conf, jit, [white])
            oi = ct.instance_pass(rays_inst, labels, iconf, jit_i)
            tr.main_pass(batch0, jitter=jit.to(DEV), white_bg=white)
            tr.instance_pass(ibatch, jitter=jit_i.to(DEV))
            if step % every == every - 1 or step == steps - 1:
                p_cpu, p_gpu = psnr(oc["rgb"], rgbs), psnr(tr.last_outputs[0].cpu(), rgbs)
                worst = max(worst, abs(p_cpu - p_gpu))
                assert abs(p_cpu - p_gpu) < 0.1, (step, p_cpu, p_gpu)
                rel_close(tr.losses[1], oc["loss_sem"], 5e-2, what=f"bf16 step {step} loss_sem")
                rel_close(tr.losses[3], oi["loss"], 1e-1, atol=5e-3, what=f"bf16 step {step} slow-fast loss")
        if steps >= 30:          # (the long run must also have LEARNT something: a parity of two runs that both stand still would say little)
            assert p_cpu > psnr(torch.full_like(rgbs, 0.5), rgbs) + 1.0
        print(f"bf16 vs oracle, C = 2 / 25 ids: PSNR after {steps} steps oracle {p_cpu:.3f} dB, HIP bf16 {p_gpu:.3f} dB; worst |delta| {worst:.4f} dB")
    finally:
        engine.set_mlp_precision("fp32")


def test_two_processes_sharing_the_gpu_do_not_disturb_each_other():
    """Two processes on ONE device (the situation of every two-rank test on the single-GPU box) each repeat the fp32x6 layer kernels, the
    exact 256- and 128-wide layer kernels, the bf16 layer kernel, the weight-gradient kernels (streamed and generating) and the fused
    first-two-layers backward, and an unrelated elementwise kernel on fixed inputs: every repeat bit-identical to the first (the kernels
    that accumulate with atomics: within 1e-4 of it) -- up to 0.2 % of the repeats may differ: two processes on one device have a low base rate of
    wrong values whatever the kernels (profiles/r03_x6_notes.txt), the kernels this test exists to catch disturbed their neighbours in 3 - 40 %.  Regression test: the first fp32x6 layer kernel (4 waves, one per SIMD) and an fp32x6 weight-gradient kernel of the same build made the
    elementwise kernel running beside them return corrupted lanes 48..63 under exactly this sharing (profiles/r03_x6_notes.txt); both are gone."""
    import subprocess, sys
    from conftest import REPO
    cmd = [sys.executable, os.path.join(REPO, "tools", "shared_gpu_stress.py")]
    procs = [subprocess.Popen(cmd + [str(m), tag, "1500"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for m, tag in ((100000, "A"), (70000, "B"))]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
