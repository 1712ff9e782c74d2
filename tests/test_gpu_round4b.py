"""Round 4, second half: the appearance head's front end as one launch each way (ABI 16).

Reference: model/radiance_field/tensoRF.py:127-137 (plane x line products, basis Linear) and :400-418 (MLP input assembly with the
positional encodings); model/renderer/panopli_tensoRF_renderer.py:103-110 (the call site on the active samples).
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, _import, build_model, scene

pytestmark = pytest.mark.gpu


def _render_ctx(fused, n_rays=700, res=(40, 48, 56), grad=True, cap=None, seed=31, heads=("app", "sem"), mode="softmax"):
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    C_, E = 5, 3
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P, rays, rng = scene(op, orays, seed, res, C_, E, n_rays, amp=2.2, sg=0.4)
    jitter = torch.from_numpy(rng.uniform(0, 1, n_rays).astype(np.float32))
    m = build_model(cl, P, res, C_, E, -3.0, mode)
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode=mode).to(DEV)
    prev = engine.APP_FRONT_FUSED
    engine.APP_FRONT_FUSED = fused
    try:
        out, ctx = engine.render_forward(m, r, rays.to(DEV), jitter.to(DEV), False, grad_heads=heads if grad else (), cap=cap)
        torch.cuda.synchronize()
    finally:
        engine.APP_FRONT_FUSED = prev
        engine.reset_rows_limit(torch.device(DEV, torch.cuda.current_device()))
    return m, out, ctx


@pytest.mark.parametrize("n_rays", [3, 700, 2500])
def test_app_front_fwd_against_the_three_launches_and_fp64(n_rays):
    """clift_app_front_fwd = gather + basis Linear + input encoding: positions and products bit-identical to clift_app_gather_fwd, features within
    fp32 round-off of float64 (fp32 matrix cores, two k halves; the unfused GEMM is held to the same bound), the encoded rows within the sine's
    conditioning of that, pad columns zero, rendered colours equal to 1e-5."""
    m, o_f, c_f = _render_ctx(True, n_rays)
    _, o_u, c_u = _render_ctx(False, n_rays)
    M = c_f.M
    assert M == c_u.M and M > 0 and (n_rays < 100 or M % 64 != 0)
    assert torch.equal(c_f.xa, c_u.xa)
    assert torch.equal(c_f.F, c_u.F)
    Wb = m.named_views()["appearance_basis_mat.weight"]
    F64 = c_u.F.double().cpu()
    ref = F64 @ Wb.double().cpu().t()
    scale = (F64.abs() @ Wb.double().cpu().abs().t()).clamp_min(1e-30)
    nf = Wb.shape[0]
    for c in (c_f, c_u):
        err = ((c.feat[:, :nf].double().cpu() - ref).abs() / scale).max()
        assert float(err) <= 1e-6, float(err)
    assert bool((c_f.feat[:, nf:] == 0).all())
    assert c_f.X.shape == c_u.X.shape and c_f.ldx == c_u.ldx
    # |d sin(2 f)| <= 2 |d f|: the encodings differ by at most twice the features' difference (+ one rounding)
    df = float((c_f.feat[:, :nf] - c_u.feat[:, :nf]).abs().max())
    assert float((c_f.X - c_u.X).abs().max()) <= 2 * df + 2e-7
    b5 = nf + 3 + 4 * nf + 12
    assert bool((c_f.X[:, b5:] == 0).all())
    assert float((o_f["rgb"] - o_u["rgb"]).abs().max()) <= 1e-5
    assert torch.equal(o_f["depth"], o_u["depth"])


def test_app_front_fwd_without_products_row_limit_and_bad_arguments():
    """No backward wanted: the products are not written (ctx.F is None) and the colours are the same; under a device-side row limit (sync-free
    step: buffers sized by a capacity) rows past the true count are not touched; argument checks raise."""
    from contrastive_lift_amd import _lib, engine
    from contrastive_lift_amd._lib import call, ptr, stream
    import ctypes as C
    m, o_g, c_g = _render_ctx(True, 700, grad=True)
    _, o_n, c_n = _render_ctx(True, 700, grad=False)
    assert c_n.F is None and c_g.F is not None
    assert torch.equal(o_g["rgb"], o_n["rgb"])
    # capped: same rows, the tail of the capacity-sized buffers untouched
    cap = c_g.M + 777
    _, o_c, c_c = _render_ctx(True, 700, grad=True, cap=cap)
    assert torch.equal(o_c["rgb"], o_g["rgb"])
    assert torch.equal(c_c.X[:c_g.M], c_g.X) and torch.equal(c_c.feat[:c_g.M], c_g.feat) and torch.equal(c_c.F[:c_g.M], c_g.F)
    # direct call on poisoned buffers under a limit
    views = m.named_views()
    va = engine.vm_struct(views, "appearance", c_g.res)
    Wb = views["appearance_basis_mat.weight"]
    nf, ldx = Wb.shape[0], c_g.ldx
    Mtrue = c_g.M
    X = torch.full((cap, ldx), -3.0, device=DEV); feat = torch.full((cap, 28), -3.0, device=DEV); xa = torch.full((cap, 4), -3.0, device=DEV)
    act = torch.zeros(cap, dtype=torch.int32, device=DEV); act[:Mtrue] = c_g.act_idx[:Mtrue]
    lim = engine.rows_limit(X.device)
    lim[0:1].fill_(Mtrue)
    try:
        call("clift_app_front_fwd", C.byref(c_g.ms), C.byref(va), ptr(c_g.rays), ptr(c_g.jitter), ptr(act), cap, ptr(Wb), engine._pitch(Wb), nf,
             m.pe_feat, m.pe_view, ptr(xa), ptr(feat), 28, ptr(X), ldx, None, stream())
        torch.cuda.synchronize()
    finally:
        engine.reset_rows_limit(X.device)
    assert bool((X[Mtrue:] == -3.0).all()) and bool((feat[Mtrue:] == -3.0).all()) and bool((xa[Mtrue:] == -3.0).all())
    assert torch.equal(X[:Mtrue], c_g.X) and torch.equal(xa[:Mtrue], c_g.xa)
    with pytest.raises(_lib.CliftError, match="n_features"):
        call("clift_app_front_fwd", C.byref(c_g.ms), C.byref(va), ptr(c_g.rays), ptr(c_g.jitter), ptr(act), 10, ptr(Wb), engine._pitch(Wb), 29,
             m.pe_feat, m.pe_view, ptr(xa), ptr(feat), 28, ptr(X), ldx, None, stream())
    with pytest.raises(_lib.CliftError, match="ldx"):
        call("clift_app_front_fwd", C.byref(c_g.ms), C.byref(va), ptr(c_g.rays), ptr(c_g.jitter), ptr(act), 10, ptr(Wb), engine._pitch(Wb), nf,
             m.pe_feat, m.pe_view, ptr(xa), ptr(feat), 28, ptr(X), 100, None, stream())


def _app_backward(n_rays=700, res=(40, 48, 56), seed=31):
    """Gradients of sum(rgb * cot) through the appearance head."""
    from contrastive_lift_amd import engine
    m, out, ctx = _render_ctx(True, n_rays, res, seed=seed)
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(out["rgb"].shape, generator=g).to(DEV)
    m.zero_grad_arena()
    gv = m.named_grad_views()
    engine.render_backward(m, ctx, gv, g_rgb=cot, density_grad=False)
    torch.cuda.synchronize()
    keys = ["appearance_basis_mat.weight"] + [f"appearance_plane.{i}" for i in range(3)] + [f"appearance_line.{i}" for i in range(3)] + \
           ["render_appearance_mlp.mlp.0.weight", "render_appearance_mlp.mlp.4.weight", "render_appearance_mlp.mlp.4.bias", "render_appearance_mlp.mlp.2.weight"]
    return m, ctx, cot, {k: gv[k].detach().clone() for k in keys}


@pytest.mark.parametrize("M,no,ldd,nh", [(4096, 3, 4, 128), (4099, 3, 4, 128), (249003, 3, 4, 128), (33, 3, 4, 128), (5000, 22, 24, 32), (8191, 6, 8, 224)])
def test_output_layer_backward_in_one_pass_over_a_narrower_hidden_layer(M, no, ldd, nh):
    """clift_out_layer_bwd_nh (the appearance head: 128 hidden units, 3 outputs): dX = (H > 0) . (dOut W), gW += dOut^T H, gb += column sums of
    dOut from ONE pass over H, against fp64 -- accumulating onto existing gW / gb, ragged row counts, padded pitches whose pad columns stay
    untouched (the waves past the hidden width must neither store nor add)."""
    from conftest import rel_close
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + no + nh)
    dOut = torch.zeros((M, ldd))
    dOut[:, :no] = torch.randn((M, no), generator=g)
    W = torch.randn((no, nh), generator=g) * 0.1
    H = torch.relu(torch.randn((M, nh), generator=g))
    ldh, ldw = nh + 4, nh + 8
    Hd = torch.full((M, ldh), float("nan"), device=DEV); Hd[:, :nh] = H.to(DEV)
    Wd = torch.full((no, ldw), float("nan"), device=DEV); Wd[:, :nh] = W.to(DEV)
    dX = torch.full((M, ldh), -7.0, device=DEV)
    gW = torch.full((no, ldw), 0.25, device=DEV)
    gb = torch.full((no,), -1.0, device=DEV)
    engine.call("clift_out_layer_bwd_nh", engine.ptr(dOut.to(DEV)), ldd, no, engine.ptr(Wd), ldw, engine.ptr(Hd), ldh, nh, M, engine.ptr(dX), ldh,
                engine.ptr(gW), ldw, engine.ptr(gb), 0, engine.stream())
    torch.cuda.synchronize()
    ref_dx = (dOut[:, :no].double() @ W.double()) * (H > 0)
    ref_w = dOut[:, :no].double().T @ H.double() + 0.25
    ref_b = dOut[:, :no].double().sum(0) - 1.0
    rel_close(dX[:, :nh], ref_dx, 2e-5, atol=2e-5 * float(ref_dx.abs().max()), what="fused dX")
    rel_close(gW[:, :nh], ref_w, 2e-5, atol=2e-5 * float(ref_w.abs().max()), what="fused gW")
    rel_close(gb, ref_b, 2e-5, atol=2e-5 * M ** 0.5, what="fused gb")
    assert bool((dX[:, nh:] == -7.0).all()) and bool((gW[:, nh:] == 0.25).all())


def test_appearance_output_layer_backward_fused_matches_the_two_passes():
    """The appearance chain with clift_out_layer_bwd_nh against the weight-gradient + masked-dgrad pair, through render_backward."""
    from conftest import grad_close
    from contrastive_lift_amd import engine
    outs = []
    for fused in (True, False):
        prev = engine.APP_OUT_BWD_FUSED
        engine.APP_OUT_BWD_FUSED = fused
        try:
            outs.append(_app_backward(2500)[3])
        finally:
            engine.APP_OUT_BWD_FUSED = prev
    for k in outs[0]:
        grad_close(outs[0][k].cpu(), outs[1][k].cpu(), what=k, rtol=1e-4, scale_atol=2e-6, outlier_frac=0.0, outlier_cap=1e-4)


@pytest.mark.parametrize("mode", ["softmax", "none"])
@pytest.mark.parametrize("n_rays", [5, 1500])
def test_composite_backward_with_the_output_activations_folded_in(mode, n_rays):
    """clift_composite_bwd_act against clift_composite_bwd + one clift_rows_act_bwd per head: every parameter gradient of a backward through all
    four heads (colours, semantics, fast and slow instance halves) -- sigmoid and identity rows are the same arithmetic (the weight gradients
    differ only by the order of their atomics), the softmax row's dot product is summed in class order instead of a shuffle tree."""
    from conftest import grad_close
    from contrastive_lift_amd import engine
    res = []
    for fused in (True, False):
        m, out, ctx = _render_ctx(True, n_rays, heads=("app", "sem", "fast", "slow"), mode=mode)
        g = torch.Generator().manual_seed(9)
        cots = [torch.randn(out[k].shape, generator=g).to(DEV) for k in ("rgb", "semantics", "instances")]
        m.zero_grad_arena()
        gv = m.named_grad_views()
        prev = engine.COMPOSITE_ACT_FUSED
        engine.COMPOSITE_ACT_FUSED = fused
        try:
            engine.render_backward(m, ctx, gv, g_rgb=cots[0], g_sem=cots[1], g_inst=cots[2], density_grad=True, slow_grad=True)
            torch.cuda.synchronize()
        finally:
            engine.COMPOSITE_ACT_FUSED = prev
        res.append({k: v.detach().clone() for k, v in gv.items()})
    assert float(res[0]["render_instance_mlp.slow_mlp.2.weight"].abs().max()) > 0 and float(res[0]["render_semantic_mlp.mlp.2.weight"].abs().max()) > 0
    for k in res[0]:
        grad_close(res[0][k].cpu(), res[1][k].cpu(), what=k, rtol=1e-4, scale_atol=3e-6, outlier_frac=0.0, outlier_cap=1e-4)


@pytest.mark.parametrize("res,n_rays", [((40, 48, 56), 700), ((128, 128, 128), 2049), ((24, 28, 32), 3)])
def test_density_forward_one_wave_per_ray_is_the_per_thread_kernel_bit_for_bit(res, n_rays, monkeypatch):
    """clift_density_fwd, wave-per-ray form (tap records through LDS, ray set-up once) against the per-thread form it replaced: identical sigma
    (same taps, same FMA order, same quad sum), incl. rays that miss the box and partial last sweeps (S not a multiple of 64)."""
    import ctypes as C
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P, rays, rng = scene(op, orays, 77, res, 5, 3, n_rays, amp=2.2, sg=0.4)
    rays = rays.clone()
    rays[0, 3:6] = torch.tensor([0.0, 1.0, 0.0]); rays[0, 0:3] = torch.tensor([5.0, -0.5, 5.0])        # a ray that never meets the box
    jitter = torch.from_numpy(rng.uniform(0, 1, n_rays).astype(np.float32)).to(DEV)
    m = build_model(cl, P, res, 5, 3, -3.0, "softmax")
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    rd = rays.to(DEV)
    ms = engine.march_struct(r, m)
    views = m.named_views()
    vd = engine.vm_struct(views, "density", engine.grid_res(views))
    S = int(r.n_samples)
    assert S % 64 != 0
    out = []
    for form in ("ray", "thread"):
        monkeypatch.setenv("CLIFT_DENS_FWD", form)
        sg = torch.full((n_rays, S), -1.0, device=DEV)
        call("clift_density_fwd", C.byref(ms), C.byref(vd), ptr(rd), ptr(jitter), n_rays, ptr(sg), stream())
        torch.cuda.synchronize()
        out.append(sg)
    assert torch.equal(out[0], out[1])
    assert float(out[0].max()) > 0 and bool((out[0][0] == 0).all())


@pytest.mark.parametrize("M,no,ldd", [(4096, 22, 24), (4099, 3, 4), (249003, 22, 24), (8191, 6, 8)])
def test_output_layer_backward_in_one_pass_over_a_bf16_stored_hidden_layer(M, no, ldd):
    """clift_out_layer_bwd_nh(h_bf16 = 1) (bf16 mode): H and the input gradient are bf16-stored, the products fp32.  Against fp64 over the SAME
    bf16 values of H: weight / bias gradient to fp32 round-off, the input gradient to one bf16 rounding of the fp64 value; the mask is H > 0."""
    from conftest import rel_close
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + no)
    dOut = torch.zeros((M, ldd))
    dOut[:, :no] = torch.randn((M, no), generator=g)
    W = torch.randn((no, 256), generator=g) * 0.1
    Hb = torch.relu(torch.randn((M, 256), generator=g)).to(torch.bfloat16)
    Hd, Wd = Hb.to(DEV), W.to(DEV)
    dX = torch.full((M, 256), -7.0, device=DEV, dtype=torch.bfloat16)
    gW = torch.full((no, 256), 0.25, device=DEV)
    gb = torch.full((no,), -1.0, device=DEV)
    engine.call("clift_out_layer_bwd_nh", engine.ptr(dOut.to(DEV)), ldd, no, engine.ptr(Wd), 256, engine.ptr(Hd), 256, 256, M, engine.ptr(dX), 256,
                engine.ptr(gW), 256, engine.ptr(gb), 1, engine.stream())
    torch.cuda.synchronize()
    H = Hb.double()
    ref_dx = (dOut[:, :no].double() @ W.double()) * (H > 0)
    ref_w = dOut[:, :no].double().T @ H + 0.25
    ref_b = dOut[:, :no].double().sum(0) - 1.0
    rel_close(dX.float(), ref_dx, 2.0 ** -8, atol=1e-6 * float(ref_dx.abs().max()), what="fused dX (bf16-stored)")
    rel_close(gW, ref_w, 2e-5, atol=2e-5 * float(ref_w.abs().max()), what="fused gW")
    rel_close(gb, ref_b, 2e-5, atol=2e-5 * M ** 0.5, what="fused gb")
    # and the pair of launches it replaces, through the engine (bf16 mode)
    dX2 = torch.empty((M, 256), device=DEV, dtype=torch.bfloat16)
    prev = engine.set_mlp_precision("bf16")
    try:
        engine.gemm(M, 256, no, dOut.to(DEV), ldd, Wd, 256, dX2, 256, b_trans=1, mask=Hd, ldmask=256)
    finally:
        engine.set_mlp_precision(prev)
    torch.cuda.synchronize()
    if M >= 4096:
        assert torch.equal(dX, dX2)


@pytest.mark.parametrize("comps,C_", [(16, 5), (32, 5), (48, 60)])
def test_other_component_and_class_counts_against_the_oracle(comps, C_):
    """The appearance front end's other instantiations (16 / 32 components per plane: tensoRF.py:34-41 `num_appearance_comps`) and the
    compositing backward's one-thread-per-sample form (more than 48 classes), forward and every gradient against the CPU oracle."""
    from conftest import grad_close, rel_close
    from test_gpu_parity import _run_forward_backward
    cl, op, orender, ofld, olosses, orays = _import()
    res, E, n_rays = (24, 28, 32), 3, 400
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P = op.add_blob(op.make_params(41 + comps, res, C_, E, n_app=(comps,) * 3), res, 2.3, 0.42)
    _, rays, rng = scene(op, orays, 41, res, 2, E, n_rays, amp=2.3, sg=0.42)
    jitter = torch.from_numpy(rng.uniform(0, 1, n_rays).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((n_rays, 3), (n_rays, C_), (n_rays, 2 * E))]
    Pg = op.clone_params(P, requires_grad=True)
    cfg = orender.RenderCfg(aabb, res, density_shift=-3.0, semantic_weight_mode="softmax")
    o = orender.render_forward(Pg, rays, cfg, jitter, False)
    ((o[0] * cots[0]).sum() + (o[1] * cots[1]).sum() + (o[2] * cots[2]).sum()).backward()
    m = cl.TensorVMSplit(list(res), num_appearance_comps=(comps,) * 3, num_semantics_comps=(32, 32, 32), num_instance_comps=(32, 32, 32),
                         num_semantic_classes=C_, dim_feature_instance=2 * E, splus_density_shift=-3.0,
                         output_mlp_semantics=torch.nn.Softmax(dim=-1), use_semantic_mlp=True, use_instance_mlp=True, slow_fast_mode=True, device=DEV)
    missing, unexpected = m.load_state_dict({k: v.to(DEV) for k, v in P.items()}, strict=True)
    assert not missing and not unexpected
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots)
    for a, b, nm in zip(outs[:4], o[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=nm)
    for k, gr in grads.items():
        ref = Pg[k].grad
        ref = torch.zeros_like(Pg[k]) if ref is None else ref
        got = torch.zeros_like(ref) if gr is None else gr.detach().cpu()
        # (scale_atol 1e-3: on this small scene the K = 3 layers' gradients cancel heavily -- the fp32 CPU oracle itself is 7e-5 of the scale away
        # from its float64 evaluation in 29 entries of the fast instance head's first layer, the exact-fp32 HIP path 1e-4)
        grad_close(got, ref, what=f"grad {k}", rtol=2e-3, scale_atol=1e-3,
                   outlier_frac=(5e-3 if k.split(".")[0].endswith(("_plane", "_line")) else 1e-2 if k.startswith("appearance_basis") else 1e-3),
                   outlier_cap=1e-2)


@pytest.mark.parametrize("M", [4096, 4101, 62003, 249001])
def test_first2_bf16_bwd_against_the_pair_of_launches_and_fp64(M):
    """clift_xyz_head_first2_bf16_bwd (bf16 mode): the second layer's input gradient formed, rounded to bf16, masked and consumed by the K = 3 layer's
    weight / bias gradient in one launch, against (a) the masked bf16 dgrad launch + clift_linear_k3_bwd over its bf16-stored result -- the same
    products, another summation order -- and (b) float64 over the same bf16-rounded dH1."""
    from conftest import rel_close
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(M)
    d = (torch.randn(M, 256, generator=g) / 8).to(torch.bfloat16).to(DEV)
    h1 = torch.relu(torch.randn(M, 256, generator=g)).to(torch.bfloat16).to(DEV)
    W1 = (torch.randn(256, 256, generator=g) / 16).to(DEV)
    xa = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).contiguous().to(DEV)
    gW, gb = torch.full((256, 4), 0.5, device=DEV), torch.full((256,), -0.25, device=DEV)
    call("clift_xyz_head_first2_bf16_bwd", ptr(d), 256, ptr(W1), 256, ptr(h1), 256, ptr(xa), M, ptr(gW), 4, ptr(gb), stream())
    # the pair
    prev = engine.set_mlp_precision("bf16")
    try:
        dn = torch.empty(M, 256, dtype=torch.bfloat16, device=DEV)
        engine.gemm(M, 256, 256, d, 256, W1, 256, dn, 256, b_trans=1, mask=h1, ldmask=256)
        gW2, gb2 = torch.full((256, 4), 0.5, device=DEV), torch.full((256,), -0.25, device=DEV)
        call("clift_linear_k3_bwd", ptr(xa), ptr(dn), 256, M, 256, ptr(gW2), 4, ptr(gb2), 1, stream())
    finally:
        engine.set_mlp_precision(prev)
    torch.cuda.synchronize()
    dn64 = dn.double().cpu()
    ref_w = dn64.T @ xa[:, :3].double().cpu() + 0.5
    ref_b = dn64.sum(0) - 0.25
    for name, w_, b_ in (("fused", gW, gb), ("pair", gW2, gb2)):
        rel_close(w_[:, :3], ref_w, 2e-5, atol=2e-5 * float((ref_w - 0.5).abs().max()) + 1e-6, what=f"{name} gW0")
        rel_close(b_, ref_b, 2e-5, atol=2e-5 * float((ref_b + 0.25).abs().max()) + 1e-6, what=f"{name} gb0")
    assert bool((gW[:, 3] == 0.5).all())
