"""Pin the CPU oracle to outputs of the reference itself (tests/golden/*.npz, made by make_golden.py).

CPU-only.  Tolerance: the oracle and the reference are both fp32 CPU-PyTorch, so they agree to
round-off; rtol 2e-5 (explicit-tap variant: 1e-4) is asserted -- far inside the 1e-3 product tolerance.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, T, rel_close
from oracle import params as op, rays as orays, field as ofld, render as orender, losses as olosses

TIGHT = 2e-5


def test_g1_ray_generation():
    g = load_golden("g1_rays")
    H, W = int(g["H"]), int(g["W"])
    i, j = orays.pixel_grid(H, W)
    assert torch.equal(i, T(g["grid_i"])) and torch.equal(j, T(g["grid_j"]))
    K = torch.from_numpy(g["K"])
    d = orays.camera_dirs(H, W, K)
    rel_close(d, g["dirs"], TIGHT, what="dirs")
    o, dd = orays.world_rays(d.to(torch.float32), T(g["c2w"]))
    rel_close(o, g["o"], TIGHT, what="o")
    rel_close(dd, g["d"], TIGHT, what="d")
    rel_close(orays.sphere_far(o, dd), g["far"], TIGHT, what="far")


def test_sphere_far_asserts_outside_unit_sphere():
    o = torch.tensor([[3.0, 0, 0]])
    d = torch.tensor([[0.0, 1.0, 0]])
    with pytest.raises(AssertionError):
        orays.sphere_far(o, d)


def test_g2_sampling_and_host_scalars():
    g = load_golden("g2_sampling")
    cfg = orender.RenderCfg(T(g["aabb"]), tuple(int(x) for x in g["res"]))
    assert cfg.n_samples == int(g["n_samples"])
    rel_close(cfg.step_size, g["step_size"], 1e-7, what="step")
    rel_close(cfg.units, g["units"], 1e-7, what="units")
    rel_close(cfg.inv_extent2, g["inv_box_extent"], 1e-7, what="inv_extent")
    rays = T(g["rays"])
    pts, z, m = orender.sample_along_rays(rays, cfg, None)
    assert torch.equal(z, T(g["z0"])) and torch.equal(pts, T(g["pts0"])) and torch.equal(m, T(g["mask0"]))
    assert torch.equal(orender.normalize(pts, cfg), T(g["xn0"]))
    pts, z, m = orender.sample_along_rays(rays, cfg, T(g["jitter"]))
    assert torch.equal(z, T(g["z1"])) and torch.equal(pts, T(g["pts1"])) and torch.equal(m, T(g["mask1"]))
    for tag in "abc":
        c2 = orender.RenderCfg(T(g[f"hs_{tag}_aabb"]), tuple(int(x) for x in g[f"hs_{tag}_grid"]),
                               step_ratio=float(g[f"hs_{tag}_ratio"]))
        assert c2.n_samples == int(g[f"hs_{tag}_n_samples"]), tag
        rel_close(c2.step_size, g[f"hs_{tag}_step_size"], 1e-7, what="hs step")


@pytest.mark.parametrize("explicit", [False, True])
def test_g3_field(explicit):
    g = load_golden("g3_field")
    res = tuple(int(x) for x in g["res"])
    P = op.make_params(int(g["seed"]), res, int(g["C"]), int(g["E"]))
    xn, vd = T(g["xn"]), T(g["viewdirs"])
    tol = 1e-4 if explicit else TIGHT
    rel_close(ofld.density_raw(P, xn, -10.0, explicit), g["density_raw"], tol, what="density_raw")
    rel_close(ofld.density(P, xn, -10.0, explicit), g["density"], tol, what="density")
    feat = ofld.appearance_feature(P, xn, explicit)
    rel_close(feat, g["app_feat"], tol, what="app_feat")
    rel_close(ofld.appearance_mlp(P, vd, feat), g["rgb"], tol, what="rgb")
    rel_close(ofld.semantic_mlp(P, xn), g["sem"], tol, what="sem")
    rel_close(ofld.instance_mlp(P, xn), g["inst"], tol, what="inst")
    rel_close(ofld.posenc(torch.tensor([[1.0, 2.0, 3.0]]), 2), g["pe"], 1e-6, what="pe")


def test_g5_alpha():
    g = load_golden("g5_alpha")
    a, w, bg = orender.sigma_to_weights(T(g["sigma"]), T(g["dist"]))
    assert torch.equal(a, T(g["alpha"])) and torch.equal(w, T(g["weight"])) and torch.equal(bg, T(g["bg"]))


def _digest_check(g, prefix, grads, tol=1e-4, stride=17):
    n = 0
    for k, gr in grads.items():
        key = f"{prefix}sub.{k}"
        if key not in g:
            continue
        flat = torch.zeros(1) if gr is None else gr.detach().reshape(-1)
        sub = flat if flat.numel() <= 4096 else flat[::stride]
        rel_close(flat.norm(), g[f"{prefix}norm.{k}"], tol, atol=1e-9, what=f"{prefix}norm.{k}")
        rel_close(sub, g[key], tol, what=key)
        n += 1
    assert n > 0
    return n


@pytest.mark.parametrize("mode", ["softmax", "none", "argmax"])
@pytest.mark.parametrize("white", [False, True])
def test_g6_forward_and_grads(mode, white):
    g = load_golden("g6a_forward_argmax" if mode == "argmax" else "g6_forward")
    res = tuple(int(x) for x in g["res"])
    C, E = int(g["C"]), int(g["E"])
    P = op.clone_params(op.add_blob(op.make_params(int(g["seed"]), res, C, E), res, 2.5, 0.45), requires_grad=True)
    cfg = orender.RenderCfg(T(g["aabb"]), res, density_shift=float(g["shift"]), semantic_weight_mode=mode)
    tag = f"{mode}_{'w' if white else 'b'}"
    rgb, sem, inst, depth, feats, dreg = orender.render_forward(P, T(g["rays"]), cfg, T(g["jitter"]), white)
    rel_close(rgb, g[f"{tag}.rgb"], TIGHT, what="rgb")
    rel_close(sem, g[f"{tag}.sem"], TIGHT, what="sem")
    rel_close(inst, g[f"{tag}.inst"], TIGHT, what="inst")
    rel_close(depth, g[f"{tag}.depth"], TIGHT, what="depth")
    assert tuple(feats.shape) == (1, 1)
    rel_close(dreg, g[f"unpinned_{tag}.dist_reg"], TIGHT, what="dist_reg (unpinned: same restated formula)")
    L = (rgb * T(g["cot_rgb"])).sum() + (sem * T(g["cot_sem"])).sum() + (inst * T(g["cot_inst"])).sum()
    L.backward()
    _digest_check(g, f"{tag}.g", {k: v.grad for k, v in P.items()})


def test_g7_instance_and_segment_feature():
    g = load_golden("g7_instance_segment")
    res = tuple(int(x) for x in g["res"])
    C, E = int(g["C"]), int(g["E"])
    P = op.clone_params(op.add_blob(op.make_params(int(g["seed"]), res, C, E), res, 2.5, 0.45), requires_grad=True)
    cfg = orender.RenderCfg(T(g["aabb"]), res, density_shift=float(g["shift"]))
    inst, xyz = orender.render_instance_feature(P, T(g["rays"]), cfg)
    rel_close(inst, g["inst"], TIGHT, what="inst")
    rel_close(xyz, g["xyz"], TIGHT, what="xyz")
    (inst * T(g["cot_inst"])).sum().backward()
    _digest_check(g, "inst.g", {k: v.grad for k, v in P.items()})
    for v in P.values():
        v.grad = None
    seg = orender.render_segment_feature(P, T(g["rays"]), cfg)
    rel_close(seg, g["seg"], TIGHT, what="seg")
    (seg * T(g["cot_seg"])).sum().backward()
    _digest_check(g, "seg.g", {k: v.grad for k, v in P.items()})


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g19_grid_heads(tag):
    """Semantic / instance heads on their own VM grids (tensoRF.py:70-83,142-156; golden G19 = the reference's forward, instance-feature and
    segment-feature passes with the gradients of every parameter, and its TV term with the grid terms on): (a) both heads on grids, (b) semantic
    MLP + instance grid with the slow-fast twin."""
    g = load_golden("g19_grid_heads")
    res = tuple(int(x) for x in g["res"])
    C, E = int(g["C"]), int(g["E"])
    sem_grid, inst_grid, sf = bool(int(g[f"{tag}.sem_grid"])), bool(int(g[f"{tag}.inst_grid"])), bool(int(g[f"{tag}.slow_fast"]))
    P = op.clone_params(op.add_blob(op.make_params(int(g["seed"]), res, C, E, slow_fast=sf, sem_grid=sem_grid, inst_grid=inst_grid), res, 2.5, 0.45),
                        requires_grad=True)
    cfg = orender.RenderCfg(T(g["aabb"]), res, density_shift=float(g["shift"]))
    rays = T(g[f"{tag}.rays"])
    rgb, sem, inst, depth, feats, dreg = orender.render_forward(P, rays, cfg, T(g[f"{tag}.jitter"]), False)
    for nm, a in (("rgb", rgb), ("sem", sem), ("inst", inst), ("depth", depth)):
        rel_close(a, g[f"{tag}.{nm}"], TIGHT, what=nm)
    ((rgb * T(g[f"{tag}.cot_rgb"])).sum() + (sem * T(g[f"{tag}.cot_sem"])).sum() + (inst * T(g[f"{tag}.cot_inst"])).sum()).backward()
    _digest_check(g, f"{tag}.g", {k: v.grad for k, v in P.items()})
    for v in P.values():
        v.grad = None
    fi, xyz = orender.render_instance_feature(P, rays, cfg)
    rel_close(fi, g[f"{tag}.f_inst"], TIGHT, what="instance features")
    rel_close(xyz, g[f"{tag}.f_xyz"], TIGHT, what="surface points")
    (fi * T(g[f"{tag}.cot_inst"])).sum().backward()
    _digest_check(g, f"{tag}.fi.g", {k: v.grad for k, v in P.items()})
    for v in P.values():
        v.grad = None
    fs = orender.render_segment_feature(P, rays, cfg)
    rel_close(fs, g[f"{tag}.f_seg"], TIGHT, what="segment features")
    (fs * T(g[f"{tag}.cot_sem"])).sum().backward()
    _digest_check(g, f"{tag}.fs.g", {k: v.grad for k, v in P.items()})
    for v in P.values():
        v.grad = None
    tv = olosses.total_tv(P)
    rel_close(tv, g[f"{tag}.tv"], TIGHT, what="TV with the grid terms")
    tv.backward()
    _digest_check(g, f"{tag}.tv.g", {k: v.grad for k, v in P.items() if k.split(".")[0].endswith(("_plane", "_line"))})


def test_g8_contrastive():
    g = load_golden("g8_losses")
    for tag in "abcd":
        f = T(g[f"con_{tag}.f"]).requires_grad_(True)
        L = olosses.contrastive(f, T(g[f"con_{tag}.y"]), 100.0)
        rel_close(L, g[f"con_{tag}.loss"], TIGHT, atol=1e-7, what=f"contrastive {tag}")
        gr = torch.autograd.grad(L, f, allow_unused=True)[0] if L.requires_grad else None
        rel_close(torch.zeros_like(f) if gr is None else gr, g[f"con_{tag}.grad"], 1e-4, what=f"contrastive grad {tag}")


def test_g8_slow_fast_and_ema():
    g = load_golden("g8_losses")
    for tag in "abcd":
        f = T(g[f"sf_{tag}.feats"]).requires_grad_(True)
        L = olosses.slow_fast(f, T(g[f"sf_{tag}.y"]), T(g[f"sf_{tag}.conf"]))
        rel_close(L, g[f"sf_{tag}.loss"], TIGHT, atol=1e-7, what=f"slow_fast {tag}")
        gr = torch.autograd.grad(L, f)[0]
        rel_close(gr, g[f"sf_{tag}.grad"], 1e-4, what=f"slow_fast grad {tag}")
        assert float(gr[:, 3:].abs().max()) == 0.0      # slow half receives no gradient
        assert float(gr[f.shape[0] // 2:].abs().max()) == 0.0   # second half of the rays = slow set only
    res = tuple(int(x) for x in g["sf_ema.res"])
    P = op.make_params(int(g["sf_ema.seed"]), res, 2, 3)
    slow = [P[k] for k in P if ".slow_mlp." in k]
    fast = [P[k.replace(".slow_mlp.", ".mlp.")] for k in P if ".slow_mlp." in k]
    olosses.ema_(slow, fast, 0.9)
    for k in [k for k in P if ".slow_mlp." in k]:
        v = P[k].reshape(-1)
        v = v if v.numel() <= 4096 else v[::17]
        rel_close(v, g["sf_ema.slow." + k.split(".slow_mlp.")[1]].reshape(-1), 1e-6, what=k)


def test_g9_tv():
    g = load_golden("g9_tv")
    res = tuple(int(x) for x in g["res"])
    P = op.clone_params(op.make_params(int(g["seed"]), res, 2, 3), requires_grad=True)
    L = olosses.tv_plane(P["density_plane.1"])
    rel_close(L, g["tv_plane1"], TIGHT, what="tv")
    rel_close(torch.autograd.grad(L, P["density_plane.1"])[0], g["tv_plane1_grad"], 1e-4, what="tv grad")
    Lt = olosses.total_tv(P, 0.1, 0.01)
    rel_close(Lt, g["total_tv"], TIGHT, what="total tv")
    Lt.backward()
    _digest_check(g, "tv.g", {k: v.grad for k, v in P.items() if v.grad is not None})


def test_g11_metrics_vs_reference():
    """psnr, robust mIoU and panoptic quality of the product's host-side metrics against the reference's own functions."""
    from contrastive_lift_amd.inference import psnr, ConfusionMatrix
    from contrastive_lift_amd.metrics import panoptic_quality
    g = load_golden("g11_metrics")
    rel_close(psnr(T(g["psnr_a"]), T(g["psnr_b"])), g["psnr"], 1e-6, what="psnr")
    cm = ConfusionMatrix(6, ignore_class=[0])
    rel_close(cm.add_batch(g["cm_pred"], g["cm_gt"], return_miou=True), g["cm_batch_miou"], 1e-9, what="batch miou")
    rel_close(cm.get_miou(), g["cm_miou"], 1e-9, what="miou")
    for k in range(6):
        pq, sq, rq = panoptic_quality(T(g[f"pq{k}.preds"]), T(g[f"pq{k}.target"]), {1, 2}, {0, 3}, allow_unknown_preds_category=True)
        # the reference divides integer tensors, i.e. each IoU is rounded to fp32 before the fp64 sum -> 1e-6
        rel_close(torch.stack([pq, sq, rq]), g[f"pq{k}.out"], 1e-6, atol=1e-9, what=f"pq case {k}")
    with pytest.raises(ValueError):
        panoptic_quality(T(g["pq0.preds"]), T(g["pq0.target"]), {1, 2}, {0, 3}, allow_unknown_preds_category=False)


@pytest.mark.parametrize("fixture", ["g12_training_steps", "g12c_training_steps_contrastive", "g12s_training_steps_segments",
                                     "g12e_training_steps_sce", "g12l_training_steps_linear_assignment", "g12g_training_steps_grid_heads",
                                     "g12gs_training_steps_grid_heads_slow_fast", "g12a_training_steps_argmax",
                                     "g12n_training_steps_nottaconf", "g12p_training_steps_noconf"])
def test_g12_three_reference_training_steps(fixture):
    """The oracle's CpuTrainer replays three training_step()s of the REFERENCE TensoRFTrainer (optimizer groups, chunked
    forwards, masked MSE + TV + confidence-weighted CE + ramped dist-reg, Adam; EMA -> slow-fast loss -> Adam on the fast
    net) with the recorded jitter / white-background draws: losses to 1e-4, every parameter after every step to a small
    fraction of one Adam step (Adam normalises the gradient, so a round-off-level gradient difference on a near-zero
    gradient moves a weight by up to lr; what is asserted is the norm to 1e-4 and elementwise 5 % of lr)."""
    from oracle.train_step import CpuTrainer
    g = load_golden(fixture)          # the second fixture: instance_loss_mode "contrastive" with use_delta (T:243-250), single MLP
    res = tuple(int(x) for x in g["res"])
    C, E = int(g["C"]), int(g["E"])
    mode = str(g["mode"]) if "mode" in g else "slow_fast"
    grids = "grid_heads" in fixture               # sixth fixture: both heads on VM grids (the allgrid overlay)
    P = op.add_blob(op.make_params(int(g["seed"]), res, C, E, slow_fast=(mode == "slow_fast"), sem_grid=grids, inst_grid=grids), res, 2.5, 0.45)
    cfg = orender.RenderCfg(T(g["aabb"]), res, density_shift=float(g["shift"]),
                            semantic_weight_mode=str(g["weight_mode"]) if "weight_mode" in g else "softmax")      # eighth fixture: "argmax" (R:142-143)
    tr = CpuTrainer(P, cfg, chunk=int(g["chunk"]), epoch=int(g["epoch"]), class_weights=T(g["class_weights"]), late_semantic_optimization=1,
                    instance_optimization_epoch=3,
                    instance_loss_mode=mode, use_delta=bool(int(g["use_delta"])) if "use_delta" in g else False,
                    sce=(tuple(float(x) for x in g["sce"]) if "sce" in g and float(g["sce"][1]) != 0.0 else None))   # 4th fixture: SCELoss
    rel_close(tr.l_dist, g["lambda_dist"], 1e-6, what="dist-reg ramp")
    # optimizer layout of the reference (T:98-103): 7 main groups (4 grid groups at 20 lr, 3 net groups at lr) + 1 instance group
    og = g["opt_groups"]
    nmain = int(g["opt_group_counts"][0])
    assert list(g["opt_group_counts"]) == ([10, 4] if grids else [7, 1])          # grid heads: + semantic planes / lines / basis (+ MLP in place); instance planes, lines, basis, MLP
    main_lr = {float(x["lr"]): sum(p.numel() for p in x["params"]) for x in tr.opt_main.param_groups}
    want = {}
    for row in og[:nmain]:
        want[float(row[0])] = want.get(float(row[0]), 0) + int(row[4])
    assert main_lr == want
    inst_lr = {float(x["lr"]): sum(p.numel() for p in x["params"]) for x in tr.opt_inst.param_groups}
    want = {}
    for row in og[nmain:]:
        want[float(row[0])] = want.get(float(row[0]), 0) + int(row[4])
    assert inst_lr == want
    assert tr.opt_main.param_groups[0]["betas"] == (0.9, 0.99) and tr.opt_inst.param_groups[0]["betas"] == (0.9, 0.999)
    assert all(abs(x["weight_decay"] - og[0, 1]) < 1e-20 for x in tr.opt_main.param_groups + tr.opt_inst.param_groups)
    for st in range(int(g["steps"])):
        seg = None
        if f"s{st}.srays" in g:          # third fixture: the segment-consistency term of T:185-197 (batch[2])
            seg = dict(rays=T(g[f"s{st}.srays"]), group=torch.from_numpy(g[f"s{st}.sgroup"]), conf=T(g[f"s{st}.sconf"]),
                       jitter=T(g[f"s{st}.sjitter"]), n_groups=6)
        ce_mode = str(g["ce_mode"]) if "ce_mode" in g else "TTAConf"          # ninth / tenth fixtures: probabilistic_ce_mode NoTTAConf / NoConf (T:177-182)
        o = tr.main_pass(T(g[f"s{st}.rays"]), T(g[f"s{st}.rgbs"]), T(g[f"s{st}.probs"]), T(g[f"s{st}.confs"]), T(g[f"s{st}.jitter"]),
                         [bool(x) for x in g[f"s{st}.white"]], mask=torch.from_numpy(g[f"s{st}.mask"]), segments=seg, ce_mode=ce_mode,
                         semantics=T(g[f"s{st}.probs"]).argmax(-1))           # (the generator's label map: batch[0]["semantics"] = probs.argmax(-1))
        if seg is not None:
            rel_close(o["loss_segment"], g[f"s{st}.loss_segment"], 1e-4, what=f"step {st} loss_segment")
        rel_close(o["loss_rgb"], g[f"s{st}.loss_rgb"], 1e-4, what=f"step {st} loss_rgb")
        rel_close(o["loss_sem"], g[f"s{st}.loss_sem"], 1e-4, what=f"step {st} loss_sem")
        oi = tr.instance_pass(T(g[f"s{st}.irays"]), torch.from_numpy(g[f"s{st}.labels"]), T(g[f"s{st}.iconf"]), T(g[f"s{st}.ijitter"]))
        rel_close(oi["loss"], g[f"s{st}.loss_clustering"], 1e-4, what=f"step {st} loss_clustering")
        for k, v in tr.P.items():
            flat = v.detach().reshape(-1)
            sub = flat if flat.numel() <= 4096 else flat[::17]
            lr = 1e-2 if k.split(".")[0].endswith(("_plane", "_line")) else 5e-4
            rel_close(flat.norm(), g[f"s{st}.pnorm.{k}"], 1e-4, atol=1e-7, what=f"step {st} |{k}|")
            assert float((sub - T(g[f"s{st}.psub.{k}"]).reshape(-1)).abs().max()) <= 0.05 * lr * (st + 1), (st, k)


def test_g13_postprocess_host_parts():
    """create_instances_from_semantics and distance_to_depth of the product (pure torch, device-agnostic) against the
    reference's outputs; assign_clusters (nearest-centroid kernel) is checked on the GPU (tests/test_gpu_parity.py)."""
    from contrastive_lift_amd.inference import create_instances_from_semantics, distance_to_depth
    g = load_golden("g13_postprocess")
    things = [int(x) for x in g["things"]]
    for j in range(int(g["n_img"])):
        got = create_instances_from_semantics(T(g[f"inst{j}"]), T(g[f"sem{j}"]), things)
        assert torch.equal(got, T(g[f"thing{j}"]))
    rel_close(distance_to_depth(T(g["K"]), T(g["dist"])), g["depth"], 1e-6, what="distance_to_depth")


def test_g18_sce_loss_and_semantic_weights():
    """oracle.losses.sce_rows / semantic_weights against the reference's SCELoss / get_semantic_weights (loss.py:29-59)."""
    from oracle import losses as olosses
    g = load_golden("g18_sce")
    assert torch.equal(olosses.semantic_weights(False, g["w.fg_idx"], 7), T(g["w.plain"]))
    assert torch.equal(olosses.semantic_weights(True, g["w.fg_idx"], 7), T(g["w.fg"]))
    for tag in "abcd":
        pred = T(g[f"{tag}.pred"]).requires_grad_(True)
        a, b = (float(x) for x in g[f"{tag}.ab"])
        rows = olosses.sce_rows(pred, T(g[f"{tag}.p"]), T(g[f"{tag}.w"]), a, b)
        rel_close(rows, g[f"{tag}.rows"], 1e-5, atol=1e-6, what=f"sce rows {tag}")
        gr = torch.autograd.grad((rows * T(g[f"{tag}.conf"])).mean(), pred)[0]
        rel_close(gr, g[f"{tag}.grad"], 1e-4, atol=1e-8, what=f"sce grad {tag}")


def _g21_check_params(g, tag, e, st, named, rtol, lr_frac, steps_done):
    for k, v in named.items():
        flat = v.detach().cpu().reshape(-1)
        sub = flat if flat.numel() <= 4096 else flat[::int(g["stride"])]
        lr = 1e-2 if k.split(".")[0].endswith(("_plane", "_line")) else 5e-4
        rel_close(flat.norm(), g[f"{tag}.e{e}.s{st}.pnorm.{k}"], rtol, atol=1e-6, what=f"{tag} e{e} s{st} |{k}|")
        want = T(g[f"{tag}.e{e}.s{st}.psub.{k}"]).reshape(-1)
        assert sub.shape == want.shape, (tag, e, st, k, sub.shape, want.shape)
        diff = float((sub - want).abs().max())
        assert diff <= lr_frac * lr * steps_done + 1e-7, f"{tag} e{e} s{st} {k}: max |diff| {diff:.3e} (lr {lr})"


@pytest.mark.parametrize("tag", ["A", "B"])
def test_g21_epoch_boundary(tag):
    """The oracle's CpuTrainer + oracle/grid_ops.py replay the REFERENCE trainer's composed epoch boundary (golden G21: on_train_epoch_start =
    ramp -> alpha-mask shrink -> upsample -> weight_decay 0 -> optimizer rebuild; scheduler step at the last batch): renderer state after
    every hook exactly, losses to 1e-4, every parameter after every step to a fraction of an Adam step.  Scenario B: a shrink in an epoch with
    no upsample leaves the optimizer on the replaced table tensors (the reference's behaviour: cropped tables frozen until the next rebuild)."""
    from oracle.train_step import CpuTrainer
    g = load_golden("g21_epoch_boundary")
    res = tuple(int(x) for x in g["res"])
    C, E = int(g["C"]), int(g["E"])
    P = op.add_blob(op.make_params(int(g["seed"]), res, C, E, grid_scale=float(g["grid_scale"])), res, 2.5, 0.3)
    cfg = orender.RenderCfg(T(g["aabb"]), res, density_shift=float(g["shift"]))
    tr = CpuTrainer(P, cfg, chunk=int(g["chunk"]), epoch=0, class_weights=T(g["class_weights"]), late_semantic_optimization=1,
                    instance_optimization_epoch=2)
    shrink_at, up_at = [int(x) for x in g[f"{tag}.bbox_aabb_reset_epochs"]], [int(x) for x in g[f"{tag}.grid_upscale_epochs"]]
    done = 0
    for e in range(int(g[f"{tag}.epochs"])):
        tr.on_train_epoch_start(e, shrink_at, up_at, min_grid_dim=res[0], max_grid_dim=16)
        assert tuple(int(x) for x in g[f"{tag}.e{e}.grid"]) == tuple(tr.cfg.grid_dim) and int(g[f"{tag}.e{e}.n_samples"]) == tr.cfg.n_samples
        rel_close(tr.cfg.aabb, g[f"{tag}.e{e}.aabb"], 1e-6, what=f"{tag} e{e} aabb")
        rel_close(tr.cfg.step_size, g[f"{tag}.e{e}.step_size"], 1e-6, what=f"{tag} e{e} step size")
        rel_close(tr.l_dist, g[f"{tag}.e{e}.lambda_dist"], 1e-6, atol=1e-12, what=f"{tag} e{e} dist-reg ramp")
        assert float(g[f"{tag}.e{e}.weight_decay"]) == float(tr.weight_decay)
        # the optimizer holds what the reference's holds: after the shrink-only hook of scenario B still the PREVIOUS tables
        want_numel = sorted(int(x) for x in g[f"{tag}.e{e}.opt_numel"])
        held = sum(p.numel() for grp in tr.opt_main.param_groups + tr.opt_inst.param_groups for p in grp["params"])
        assert held == sum(want_numel), (tag, e, held, sum(want_numel))
        for st in range(int(g[f"{tag}.steps"])):
            k = f"{tag}.e{e}.s{st}."
            o = tr.main_pass(T(g[k + "rays"]), T(g[k + "rgbs"]), T(g[k + "probs"]), T(g[k + "confs"]), T(g[k + "jitter"]),
                             [bool(x) for x in g[k + "white"]], mask=torch.from_numpy(g[k + "mask"]))
            rel_close(o["loss_rgb"], g[k + "loss_rgb"], 1e-4, what=k + "loss_rgb")
            rel_close(o["loss_sem"], g[k + "loss_sem"], 1e-4, atol=1e-9, what=k + "loss_sem")
            if e >= 2:
                oi = tr.instance_pass(T(g[k + "irays"]), torch.from_numpy(g[k + "labels"]), T(g[k + "iconf"]), T(g[k + "ijitter"]))
                rel_close(oi["loss"], g[k + "loss_clustering"], 1e-4, what=k + "loss_clustering")
            done += 1
            _g21_check_params(g, tag, e, st, tr.P, 1e-4, 0.05, done)
        tr.end_of_epoch()
    if tag == "B":          # the cropped tables did not move during the shrink-only epoch; the MLPs did
        a, b = g["B.e1.s0.psub.density_plane.0"], g["B.e1.s1.psub.density_plane.0"]
        assert float(np.abs(a - b).max()) == 0.0
        assert float(np.abs(g["B.e1.s0.psub.render_appearance_mlp.mlp.0.weight"] - g["B.e1.s1.psub.render_appearance_mlp.mlp.0.weight"]).max()) > 0


def test_g21_learning_rate_restart_unpinned():
    """Scenario C of G21 (UNPINNED: it rests on the stub of Lightning's ``strategy.setup_optimizers``, which replaces the scheduler objects):
    decay_step [1, 3] with a grid upsample at epoch 2 -- the MultiStepLR milestones count epochs since the last rebuild.  The oracle's
    schedulers and the product's ``scheduler_step`` arithmetic give the same table."""
    g = load_golden("g21_epoch_boundary")
    lr = g["unpinned_C.opt_lr"]
    decay, ups = [int(x) for x in g["unpinned_C.decay_step"]], [int(x) for x in g["unpinned_C.grid_upscale_epochs"]]
    steps = 0
    for e in range(lr.shape[0]):
        if e in ups:
            steps = 0
        scale = 0.5 ** sum(1 for m in decay if steps >= m)
        rel_close(lr[e, 0], 1e-2 * scale, 1e-9, what=f"epoch {e} grid lr")
        rel_close(lr[e, -1], 5e-4 * scale, 1e-9, what=f"epoch {e} net lr")
        steps += 1
