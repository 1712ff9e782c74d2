"""GPU parity tests added in round 2: the BASELINE configurations that round 1 left unexercised, the full-size backward,
and the optimizer behaviour before the semantic term switches on.

  * configs[3] "MOS large_corridor_500 (500 instances), 8192 rays/batch": slow-fast and contrastive losses with 500 label
    ids at B = 1024 against the oracle; an 8192-ray training step at the Messy-Rooms class count on a 128^3 grid (whole batch ==
    the reference's 2048-ray chunking, main + instance pass).
  * configs[1] at FULL size (4096 rays, 128^3 => S = 440, C = 22): outputs and EVERY parameter gradient against the CPU oracle --
    pins the >= 160 k-row weight-gradient launches, the persistent dgrad and the 128^3 scatter at the bench shape.
  * configs[4] "full-frame 1296x968 render, ray tiles sharded": one frame rendered at the halved step ratio on one GPU, and the
    same frame through ``render_rays_sharded`` with the real renderer in two ranks (gloo) sharing the GPU: bit-identical.
  * epoch < late_semantic_optimization: torch's Adam skips the semantic MLP (grad None); ours must too.
"""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import grad_close, rel_close, usable_cores, load_golden, T
from test_gpu_parity import _import, build_model, scene, _run_forward_backward

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ============================================================================ optimizer: semantic head before its loss term exists
def test_adam_skips_semantic_head_until_late_semantic_epoch():
    """Reference T:175,198: before ``late_semantic_optimization`` the semantic output is not in the loss, the head's .grad is None
    and torch.optim.Adam neither decays nor moves it, and its per-parameter step count starts when the term switches on.
    Three steps at epoch 0 (head must stay BIT-identical), then two at epoch 1, against the oracle's torch Adam."""
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    from oracle.train_step import CpuTrainer
    res, C_, E, B = (20, 24, 28), 4, 3, 320
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P, rays, rng = scene(op, orays, 55, res, C_, E, B, amp=2.3, sg=0.42)
    rgbs = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32))
    probs = torch.softmax(torch.from_numpy(rng.standard_normal((B, C_)).astype(np.float32)), -1)
    conf = torch.from_numpy(rng.uniform(0.2, 1, B).astype(np.float32))
    m = build_model(cl, P, res, C_, E, -3.0)
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    cfg = default_config(chunk=0, late_semantic_optimization=1, instance_optimization_epoch=3)
    tr = HotPathTrainer(m, r, cfg, current_epoch=0)
    ct = CpuTrainer(P, orender.RenderCfg(aabb, res, density_shift=-3.0), chunk=4096, epoch=0, late_semantic_optimization=1)
    sem0 = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("render_semantic_mlp")}
    batch0 = dict(rays=rays.to(DEV), rgbs=rgbs.to(DEV), probabilities=probs.to(DEV), confidences=conf.to(DEV), mask=None)

    def both(step):
        jit = torch.from_numpy(rng.uniform(0, 1, B).astype(np.float32))
        white = bool(step % 2)
        ct.main_pass(rays, rgbs, probs, conf, jit, [white])
        tr.main_pass(batch0, jitter=jit.to(DEV), white_bg=white)
    for step in range(3):
        both(step)
    sd = m.state_dict()
    for k, v in sem0.items():
        assert torch.equal(sd[k], v), f"{k} moved although the semantic head has no gradient source yet"
        assert torch.equal(ct.P[k].detach(), P[k]), k
    assert tr.opt_main.t == {"grids": 3, "net_app": 3, "net_sem": 0}
    # epoch 1: the term switches on; bias correction of the head starts at t = 1
    tr.current_epoch = 1
    tr.on_train_epoch_start()
    ct.epoch, ct.sem_on = 1, True
    ct.l_dist = 0.005 * (1 - np.exp(-0.25))
    for step in range(3, 5):
        both(step)
    assert tr.opt_main.t == {"grids": 5, "net_app": 5, "net_sem": 2}
    sd = m.state_dict()
    moved = 0.0
    for k, pref in ct.P.items():
        if k.startswith("render_instance_mlp"):
            continue
        lr = 1e-2 if k.split(".")[0].endswith(("_plane", "_line")) else 5e-4
        # Adam normalises: an element whose gradient is round-off-sized moves by ~lr per step in a direction the summation order
        # decides, so parameters are compared in units of lr: every element within the five steps' worst case, and -- for the head
        # this test is about -- all but 1 % of the elements within 0.2 lr
        diff = (sd[k].detach().cpu() - pref.detach()).abs()
        assert float(diff.max()) <= 2 * 5 * lr, f"param {k}: max |diff| {float(diff.max()):.3e} vs lr {lr}"
        if k.startswith("render_semantic_mlp"):
            assert float((diff > 0.2 * lr).float().mean()) <= 1e-2, f"param {k}: {int((diff > 0.2 * lr).sum())}/{diff.numel()} beyond 0.2 lr"
        if k.startswith("render_semantic_mlp"):
            moved = max(moved, float((sd[k].detach().cpu() - P[k]).abs().max()))
    assert moved > 5e-4          # ... and it did start training (two Adam steps of ~lr each)


# ============================================================================ configs[3]: 500 instance ids, 8192 rays
def test_config3_losses_with_500_instance_ids():
    """MOS large_corridor_500: the per-image instance batch (B = 1024 rays, T:212) carries up to 500 distinct ids -- most labels
    then have 1-3 rays, many appear in only one half (no slow centroid / no positive pair)."""
    cl, op, orender, ofld, olosses, orays = _import()
    rng = np.random.default_rng(500)
    B, E = 1024, 3
    for trial, spread in enumerate((0.5, 0.05)):
        f = torch.from_numpy((rng.standard_normal((B, 2 * E)) * spread).astype(np.float32))
        y = torch.from_numpy(rng.integers(1, 501, B).astype(np.int64))
        assert len(torch.unique(y)) > 400
        conf = torch.from_numpy(rng.uniform(0.1, 1, B).astype(np.float32))
        fo = f.clone().requires_grad_(True)
        Lo = olosses.slow_fast(fo, y, conf)
        go = torch.autograd.grad(Lo, fo)[0]
        fd = f.to(DEV).requires_grad_(True)
        L = cl.slow_fast_loss(fd, y.to(DEV), conf.to(DEV))
        rel_close(L, Lo.detach(), 1e-4, what=f"slow_fast 500 ids [{trial}]")
        rel_close(torch.autograd.grad(L, fd)[0], go, 1e-3, atol=1e-8, what=f"slow_fast 500 ids grad [{trial}]")
        fo2 = f[:, :E].clone().requires_grad_(True)
        Lc = olosses.contrastive(fo2, y, 100.0)
        gc = torch.autograd.grad(Lc, fo2)[0]
        fd2 = f[:, :E].contiguous().to(DEV).requires_grad_(True)
        L2 = cl.contrastive_loss(fd2, y.to(DEV), 100.0)
        rel_close(L2, Lc.detach(), 1e-4, what=f"contrastive 500 ids [{trial}]")
        rel_close(torch.autograd.grad(L2, fd2)[0], gc, 1e-3, atol=1e-8, what=f"contrastive 500 ids grad [{trial}]")


def test_config3_8192_ray_training_step():
    """8192 rays per step at the Messy-Rooms class count (C = 2), 128^3 grid, 500 instance ids in the instance image: the whole-batch
    step equals the reference's chunking (4 x 2048 rays, T:108) in losses, every gradient and the parameters after the step; the
    instance pass trains only the fast MLP and the EMA moves the slow one."""
    from contrastive_lift_amd import synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    g = torch.Generator().manual_seed(12)
    jit = torch.rand(8192, generator=g).to(DEV)
    jit_i = torch.rand(1024, generator=g).to(DEV)
    res = {}
    for chunk in (0, 2048):
        model, renderer, pool = synthetic.make_scene(grid=128, num_classes=2, max_instances=3, seed=3, device=DEV, image=256, n_cams=3)
        assert int(renderer.n_samples) == 440
        cfg = default_config(chunk=chunk, instance_optimization_epoch=0, late_semantic_optimization=0)
        tr = HotPathTrainer(model, renderer, cfg, current_epoch=4)
        batch = synthetic.make_batches(pool, 8192, 1024, 2, 500, seed=21, device=DEV)
        assert len(torch.unique(batch[1][0]["instances"])) > 150
        slow0 = model.state_dict()["render_instance_mlp.slow_mlp.2.weight"].clone()
        tr.main_pass(batch[0], jitter=jit, white_bg=False)
        grads = {k: v.detach().clone() for k, v in model.named_grad_views().items()}
        losses = tr.losses.clone()
        tr.instance_pass(batch[1], jitter=jit_i)
        gi = {k: v.detach().clone() for k, v in model.named_grad_views().items() if k.startswith("render_instance_mlp")}
        res[chunk] = (grads, losses, tr.losses.clone(), gi, model.param_flat.detach().clone(), slow0, model.state_dict())
        assert bool(torch.isfinite(tr.losses).all()) and bool(torch.isfinite(model.param_flat).all())
    (g0, l0, li0, gi0, p0, slow0, sd0), (g1, l1, li1, gi1, p1, _, _) = res[0], res[2048]
    rel_close(l1[:3], l0[:3], 1e-4, what="losses, chunk 2048 vs whole batch")
    for k in g0:
        if k.startswith("render_instance_mlp"):
            assert float(g0[k].abs().max()) == 0.0
        else:
            grad_close(g1[k], g0[k], what=f"8192 rays, chunked vs whole: {k}")
    rel_close(li1[3], li0[3], 1e-4, what="slow-fast loss")
    for k in gi0:
        if ".slow_mlp." in k:
            assert float(gi0[k].abs().max()) == 0.0
        else:
            assert float(gi0[k].abs().max()) > 0.0
    assert not torch.equal(sd0["render_instance_mlp.slow_mlp.2.weight"], slow0)       # EMA
    assert float((p1 - p0).abs().max()) <= 0.2 * 1e-2


# ============================================================================ configs[1] at full size: backward vs the oracle
def test_full_size_backward_vs_oracle():
    """4096 rays, grid 128^3 (S = 440), C = 22, E = 3, one chunk: rgb / semantics / instances / depth / dist-reg and the gradient
    of every parameter against the CPU oracle (autograd).  ~250 k active samples: the weight gradients run as the large-M
    launches, the hidden layers as persistent forward / dgrad kernels, the scatter at the bench table size."""
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    res, C_, E, N = (128, 128, 128), 22, 3, 4096
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P, rays, rng = scene(op, orays, 41, res, C_, E, N, img=64, amp=3.0, sg=0.35)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    Pg = op.clone_params(P, requires_grad=True)
    cfg = orender.RenderCfg(aabb, res, density_shift=-3.0, semantic_weight_mode="softmax")
    torch.set_num_threads(usable_cores())
    o = orender.render_forward(Pg, rays, cfg, jitter, False)
    L = (o[0] * cots[0]).sum() + (o[1] * cots[1]).sum() + (o[2] * cots[2]).sum() + 3.0 * o[5]
    L.backward()
    m = build_model(cl, P, res, C_, E, -3.0, "softmax")
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    assert int(r.n_samples) == 440
    outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [3.0])
    with torch.no_grad():
        _, ctx = engine.render_forward(m, r, rays.to(DEV), jitter.to(DEV), False)
    assert ctx.M >= 160000, ctx.M          # the large-M weight-gradient branch is the one exercised
    for a, b, nm in zip(outs[:4], o[:4], ("rgb", "sem", "inst", "depth")):
        rel_close(a, b.detach(), 1e-3, what=nm)
    rel_close(outs[5], o[5].detach(), 1e-3, what="dist_reg")
    n = 0
    for k, gr in grads.items():
        ref = Pg[k].grad
        ref = torch.zeros_like(Pg[k]) if ref is None else ref
        got = torch.zeros_like(ref) if gr is None else gr.detach().cpu()
        # all but 0.1 % of the entries within 2e-3 relative + 1e-4 of the tensor's scale; those few (a sample whose weight sits on the
        # 1e-4 activity threshold, or a hidden unit on the ReLU kink, lands on the other side: one such sample moves the 48 channels of
        # the texels it touches -- hence 0.5 % for the table gradients) within 1e-2 of the scale (observed 1e-3 .. 4e-3 depending on the
        # kernels' summation order; tools/ab_persistent.py shows the same handful of entries between two GPU kernel sets; the 27 x 144
        # basis matrix sits right behind the appearance tables and gets 1 %)
        grad_close(got, ref, what=f"full-size grad {k}", rtol=2e-3, scale_atol=1e-4,
                   outlier_frac=(5e-3 if k.split(".")[0].endswith(("_plane", "_line")) else 1e-3 if "render_" in k else 1e-2), outlier_cap=1e-2)
        n += 1
    assert n >= 38


# ============================================================================ configs[4]: 1296 x 968 frame, row tiles over ranks
FRAME_H, FRAME_W, FRAME_CHUNK = 968, 1296, 65536


def _frame_setup():
    from contrastive_lift_amd import synthetic
    from contrastive_lift_amd.rays import generate_ray_table
    model, renderer, _ = synthetic.make_scene(grid=128, num_classes=2, max_instances=3, seed=4, device=DEV, image=64, n_cams=1)
    renderer.update_step_ratio(renderer.step_ratio * 0.5)          # RP:104
    K = np.array([[1170.0, 0, 647.75], [0, 1170.0, 483.75], [0, 0, 1]], np.float32)      # ScanNet colour intrinsics, 1296 x 968
    rays = generate_ray_table(FRAME_H, FRAME_W, K, synthetic.look_at((0.55, -0.3, -0.6)), device=DEV)
    return model, renderer, rays


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _frame_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contrastive_lift_amd import inference as inf
        model, renderer, rays = _frame_setup()
        whole = inf.render_rays(model, renderer, rays, FRAME_CHUNK)
        shard = inf.render_rays_sharded(model, renderer, rays, FRAME_CHUNK)
        # rays whose values differ between the two renders, per output (0 everywhere = bit-identical)
        same = [int((a != b).reshape(a.shape[0], -1).any(1).sum()) for a, b in zip(whole, shard)]
        b = inf.tile_bounds(rays.shape[0], world)
        q.put((rank, same, [tuple(x.shape) for x in shard], b))
    finally:
        dist.destroy_process_group()


def test_config4_full_frame_render_single_gpu():
    """One 1296 x 968 frame (1,254,528 rays), is_train=False, step ratio halved (S = 880), chunked like RP:114-120."""
    from contrastive_lift_amd import inference as inf
    model, renderer, rays = _frame_setup()
    assert rays.shape[0] == 1254528 and int(renderer.n_samples) >= 879
    rgb, sem, inst, dist_ = inf.render_rays(model, renderer, rays, FRAME_CHUNK)
    assert rgb.shape == (1254528, 3) and sem.shape == (1254528, 2) and inst.shape == (1254528, 6) and dist_.shape == (1254528,)
    for t in (rgb, sem, inst, dist_):
        assert bool(torch.isfinite(t).all())
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    img = rgb.reshape(FRAME_H, FRAME_W, 3)
    assert float(img.std()) > 0.01                      # the blob is in view
    # chunk-size independence (every per-ray result is computed by the same instruction sequence whatever the chunk): bit-equal
    rows = slice(400 * FRAME_W, 420 * FRAME_W)
    again = inf.render_rays(model, renderer, rays[rows], 8192)
    for a, b in zip(again, (rgb[rows], sem[rows], inst[rows], dist_[rows])):
        assert torch.equal(a, b)
    # a sample of rays against the oracle
    cl, op, orender, ofld, olosses, orays = _import()
    pick = torch.from_numpy(np.random.default_rng(0).choice(rays.shape[0], 96, replace=False)).to(DEV)
    P = {k: v.detach().cpu() for k, v in model.export_state_dict().items()}
    cfg = orender.RenderCfg(renderer.bbox_aabb.cpu(), (128, 128, 128), density_shift=-3.0, semantic_weight_mode="softmax", step_ratio=0.25)
    with torch.no_grad():
        o = orender.render_forward(P, rays[pick].cpu(), cfg, None, False)
    rel_close(rgb[pick], o[0], 1e-3, what="frame rgb vs oracle")
    rel_close(sem[pick], o[1], 1e-3, what="frame semantics vs oracle")
    rel_close(inst[pick], o[2], 1e-3, what="frame instances vs oracle")
    rel_close(dist_[pick], o[3], 1e-3, what="frame distance vs oracle")


def test_config4_full_frame_render_sharded_two_ranks():
    """The same frame through render_rays_sharded with the REAL renderer: two ranks (gloo, sharing the one GPU of the test box) each
    render a contiguous tile of 627,264 rays, one all-gather assembles the frame; every output is compared ray by ray with the unsharded
    render on both ranks, bit for bit.  (Rounds 3 - 4 tolerated <= 4 differing rays here: an intermittent event in the exact-fp32 mode that round 5
    traced to the hand-pipelined fused output layers of that mode's kernels and removed -- profiles/r05_determinism.txt.  Strict again.)"""
    import torch.multiprocessing as mp
    torch.cuda.empty_cache()                 # the two ranks share this process's GPU: hand its cached blocks back first
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ndiff, shapes, b in res:
        assert all(n == 0 for n in ndiff), (rank, ndiff)
        assert shapes[0] == (1254528, 3) and shapes[3] == (1254528,)
        assert b == [0, 627264, 1254528]


# ============================================================================ a8: k_march_fwd/bwd against the pairwise DEFINITION
def test_march_dist_loss_matches_pairwise_definition(monkeypatch):
    """The distortion loss inside k_march_fwd and its gradient inside k_march_bwd against the O(S^2) published definition
    (tests/test_dist_loss_bruteforce.py) instead of against the oracle's own prefix-sum restatement: the oracle's render is run
    with ``dist_loss`` swapped for the brute-force pairwise sum, only the dist-reg output carries a cotangent, and the density
    table gradients (the only parameters it reaches) are compared."""
    from test_dist_loss_bruteforce import brute_force
    cl, op, orender, ofld, olosses, orays = _import()
    res, C_, E, N = (24, 28, 32), 4, 3, 200
    aabb = torch.tensor([[-0.9, -0.8, -0.7], [0.8, 0.9, 0.75]])
    P, rays, rng = scene(op, orays, 61, res, C_, E, N, amp=2.4, sg=0.45)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    monkeypatch.setattr(orender, "dist_loss", lambda w, m, d: brute_force(w, m, d).float())
    Pg = op.clone_params(P, requires_grad=True)
    cfg = orender.RenderCfg(aabb, res, density_shift=-3.0)
    o = orender.render_forward(Pg, rays, cfg, jitter, False)
    (7.0 * o[5]).backward()
    m = build_model(cl, P, res, C_, E, -3.0)
    r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
    zeros = [torch.zeros(s) for s in ((N, 3), (N, C_), (N, 2 * E))]
    outs, grads = _run_forward_backward(cl, m, r, rays, jitter, False, zeros + [7.0])
    rel_close(outs[5], o[5].detach(), 1e-4, what="dist_reg vs pairwise definition")
    assert float(o[5]) > 1e-5
    for k in [f"density_plane.{i}" for i in range(3)] + [f"density_line.{i}" for i in range(3)]:
        assert float(Pg[k].grad.abs().max()) > 0
        grad_close(grads[k].detach().cpu(), Pg[k].grad, what=f"dist-reg grad {k}", rtol=2e-3, scale_atol=1e-4, outlier_frac=1e-3, outlier_cap=1e-3)


# ============================================================================ a19: SCELoss / get_semantic_weights callables
def test_sce_loss_callable_golden_g18():
    """cl.SCELoss(alpha, beta, w)(pred, soft targets) -> per-pixel loss, and its gradient, against the reference's own SCELoss
    (golden G18: zero class weight, one-hot targets hitting the 1e-8 clamps, log-probability inputs); get_semantic_weights; and the
    plain soft-target cross entropy (the reference's default loss_semantics) against torch on the device."""
    import contrastive_lift_amd as cl
    g = load_golden("g18_sce")
    assert torch.equal(cl.get_semantic_weights(False, list(g["w.fg_idx"]), 7), T(g["w.plain"]))
    assert torch.equal(cl.get_semantic_weights(True, list(g["w.fg_idx"]), 7), T(g["w.fg"]))
    for tag in "abcd":
        pred = T(g[f"{tag}.pred"]).to(DEV).requires_grad_(True)
        a, b = (float(x) for x in g[f"{tag}.ab"])
        w = T(g[f"{tag}.w"])
        rows = cl.SCELoss(a, b, w)(pred, T(g[f"{tag}.p"]).to(DEV))
        rel_close(rows, g[f"{tag}.rows"], 1e-4, atol=1e-5, what=f"SCELoss rows {tag}")
        gr = torch.autograd.grad((rows * T(g[f"{tag}.conf"]).to(DEV)).mean(), pred)[0]
        rel_close(gr, g[f"{tag}.grad"], 1e-3, atol=1e-7, what=f"SCELoss grad {tag}")
        pred2 = T(g[f"{tag}.pred"]).to(DEV).requires_grad_(True)
        ce = cl.SoftTargetCrossEntropy(w)(pred2, T(g[f"{tag}.p"]).to(DEV))
        pr = T(g[f"{tag}.pred"]).requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(pr, T(g[f"{tag}.p"]), weight=w, reduction="none")
        rel_close(ce, ref.detach(), 1e-4, atol=1e-5, what=f"soft-target CE rows {tag}")
        rel_close(torch.autograd.grad(ce.sum(), pred2)[0], torch.autograd.grad(ref.sum(), pr)[0], 1e-3, atol=1e-6, what=f"soft-target CE grad {tag}")


# ============================================================================ 8f-1: alpha-mask bounding box on the device
def test_alpha_bbox_kernel_vs_torch_pooling():
    """clift_alpha_bbox (lattice alpha + 3^3 max-pool + threshold + index bounding box in one library call) against the same
    steps done with torch ops on the dense alpha of ``get_dense_alpha`` (clift_density_points), on an anisotropic lattice and for
    three thresholds incl. one that no voxel reaches."""
    import torch.nn.functional as F
    cl, op, orender, ofld, olosses, orays = _import()
    res = (21, 34, 27)
    P = op.add_blob(op.make_params(5, res, 2, 3), res, amplitude=2.5, sigma_g=0.3)
    m = build_model(cl, P, res, 2, 3, -3.0)
    aabb = torch.tensor([[-0.9, -0.7, -0.5], [0.8, 0.7, 0.6]])
    for thr in (0.0075, 0.3, 2.0):
        r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax", alpha_mask_threshold=thr).to(DEV)
        alpha, _ = r.get_dense_alpha(m)
        pooled = F.max_pool3d(alpha.clamp(0, 1)[None, None], kernel_size=3, padding=1, stride=1)[0, 0]
        idx = torch.nonzero(pooled >= thr)
        got = r.occupied_index_box(m)
        if idx.shape[0] == 0:
            assert got is None
            continue
        lo, hi, n, _ = got
        assert lo == idx.amin(0).tolist() and hi == idx.amax(0).tolist() and n == idx.shape[0], (thr, lo, hi, n)


# ============================================================================ K = 3 layer generated inside the second layer's kernel
@pytest.mark.parametrize("M", [1, 31, 33, 4097, 70001, 600000])
def test_xyz_head_first_two_layers_fused_is_bit_identical(M):
    """clift_xyz_head_first2_fwd (first layer generated in LDS by the persistent 256x256 kernel) against clift_linear_k3_fwd followed
    by clift_gemm: same instruction sequence per element => bit-identical h2, and h1 when requested.  600000 rows = 2344 rows per
    block: the staged positions are refilled (2048 rows per fill); 1 / 31 / 33: ragged tiles."""
    import ctypes as C
    from contrastive_lift_amd import _lib, engine
    from contrastive_lift_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(M)
    xa = torch.zeros((M, 4)); xa[:, :3] = torch.rand((M, 3), generator=g) * 2 - 1
    W0 = torch.randn((256, 4), generator=g) * 0.7; W0[:, 3] = 0
    b0 = torch.randn(256, generator=g) * 0.3
    W1 = torch.randn((256, 256), generator=g) * 0.08
    b1 = torch.randn(256, generator=g) * 0.1
    xa, W0, b0, W1, b1 = (t.to(DEV).contiguous() for t in (xa, W0, b0, W1, b1))
    h1_ref = torch.empty((M, 256), device=DEV)
    call("clift_linear_k3_fwd", ptr(xa), ptr(W0), 4, ptr(b0), M, 256, 1, ptr(h1_ref), 256, 0, stream())
    h2_ref = torch.empty((M, 256), device=DEV)
    with engine.exact_fp32():            # (the fused kernel is the exact-fp32 one: its reference is too, whatever mode is forced)
        engine.gemm(M, 256, 256, h1_ref, 256, W1, 256, h2_ref, 256, bias=b1, act=1)
    for keep in (True, False):
        h1 = torch.full((M, 256), float("nan"), device=DEV) if keep else None
        h2 = torch.full((M, 256), float("nan"), device=DEV)
        call("clift_xyz_head_first2_fwd", ptr(xa), ptr(W0), 4, ptr(b0), ptr(W1), 256, ptr(b1), M, ptr(h1), 256, ptr(h2), 256, stream())
        assert torch.equal(h2, h2_ref), (M, keep, float((h2 - h2_ref).abs().max()))
        if keep:
            assert torch.equal(h1, h1_ref)


# ============================================================================ persistent 128-wide layers (appearance MLP)
@pytest.mark.parametrize("M", [1, 63, 65, 5000, 70001])
def test_persistent_128_wide_layers(M):
    """layer_n128.hip: forward K = 160 (first appearance layer, 150 inputs zero-padded to the 160-float pitch) and K = 128 with bias +
    ReLU, masked dgrad K = 128 -- against fp64, every row of ragged ranges (64-row tiles: 1 / 63 / 65), rows beyond M and pad
    columns untouched; and the same sums as the tiled kernel up to summation order."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + 11)
    for K in (160, 128):
        A = torch.randn((M, K), generator=g)
        W = (torch.randn((128, K), generator=g) / 12).contiguous()
        if K == 160:
            A[:, 150:] = 0; W[:, 150:] = 0
        bias = torch.randn(128, generator=g)
        ref = torch.relu(A.double() @ W.double().T + bias.double())
        Ad, Wd, bd = A.to(DEV), W.to(DEV), bias.to(DEV)
        out = torch.full((M + 2, 132), -7.0, device=DEV)
        engine.gemm(M, 128, K, Ad, K, Wd, K, out, 132, bias=bd, act=1)
        rel_close(out[:M, :128], ref, 2e-5, atol=2e-5 * float(ref.abs().max()), what=f"persistent 128-wide forward K={K}")
        assert bool((out[M:] == -7.0).all()) and bool((out[:, 128:] == -7.0).all())
        os.environ["CLIFT_NO_PERSISTENT"] = "1"
        try:
            out3 = torch.zeros((M, 128), device=DEV)
            engine.gemm(M, 128, K, Ad, K, Wd, K, out3, 128, bias=bd, act=1)
        finally:
            del os.environ["CLIFT_NO_PERSISTENT"]
        scale = (A.abs().double() @ W.abs().double().T + bias.abs().double()).to(DEV)
        assert float(((out[:M, :128] - out3).abs().double() / scale).max()) <= 2e-6
    A = torch.randn((M, 128), generator=g)
    W = (torch.randn((128, 128), generator=g) / 12).contiguous()
    mask = torch.randn((M, 128), generator=g)
    dX = torch.full((M + 2, 132), -7.0, device=DEV)
    engine.gemm(M, 128, 128, A.to(DEV), 128, W.to(DEV), 128, dX, 132, b_trans=1, mask=mask.to(DEV), ldmask=128)
    refd = (A.double() @ W.double()) * (mask.double() > 0)
    rel_close(dX[:M, :128], refd, 2e-5, atol=2e-5 * float(refd.abs().max()), what="persistent 128-wide dgrad")
    assert bool((dX[M:] == -7.0).all()) and bool((dX[:, 128:] == -7.0).all())


@pytest.mark.parametrize("M", [4096, 4159, 70001, 249000])
def test_persistent_128_wide_weight_gradient(M):
    """k_wgrad_n128_stream: gW (128 x K) += dY^T X and gb += colsum(dY) for K = 128 and K = 160 (first layer: pad columns of X are
    zero, so the pad columns of gW stay zero) against fp64; accumulation into a pre-filled gW; ragged last tile (64-row tiles)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + 3)
    for K in (128, 160):
        dY = torch.randn((M, 128), generator=g) * (torch.rand((M, 128), generator=g) > 0.4)
        X = torch.randn((M, K), generator=g)
        if K == 160:
            X[:, 150:] = 0
        gW0 = torch.randn((128, K), generator=g)
        gb0 = torch.randn(128, generator=g)
        gW, gb = gW0.to(DEV).contiguous(), gb0.to(DEV)
        engine.wgrad(128, K, M, dY.to(DEV), 128, X.to(DEV), K, gW, gb)
        ref = gW0.double() + dY.double().T @ X.double()
        rel_close(gW, ref, 1e-4, atol=1e-4 * float(ref.abs().max()), what=f"128-wide wgrad K={K}")
        rel_close(gb, gb0.double() + dY.double().sum(0), 1e-4, atol=1e-4 * M ** 0.5, what="bias gradient")
        if K == 160:
            assert torch.equal(gW[:, 150:].cpu(), gW0[:, 150:])


@pytest.mark.parametrize("M", [1, 31, 33, 65, 4097, 70001])
@pytest.mark.parametrize("E", [3, 4, 1])
def test_xyz_head_last_two_layers_fused(M, E):
    """clift_xyz_head_last2_fwd (the narrow output layer applied to the last hidden layer's tile in registers, cross-wave sum through
    LDS, two-tile software pipeline) against clift_gemm x 2: the hidden activation is bit-identical (same kernel body), the E outputs
    agree to summation-order round-off (checked against fp64), with and without writing the hidden activation, into a strided output
    with a column offset; rows beyond M and other columns of the output untouched."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M * 7 + E)
    A = torch.relu(torch.randn((M, 256), generator=g))
    W = (torch.randn((256, 256), generator=g) / 16).contiguous()
    b = torch.randn(256, generator=g) * 0.2
    Wo = (torch.randn((E, 256), generator=g) / 10).contiguous()
    bo = torch.randn(E, generator=g)
    Ad, Wd, bd, Wod, bod = (t.to(DEV) for t in (A, W, b, Wo, bo))
    h_ref = torch.empty((M, 256), device=DEV)
    with engine.exact_fp32():            # (clift_xyz_head_last2_fwd is the exact-fp32 kernel: so is its reference)
        engine.gemm(M, 256, 256, Ad, 256, Wd, 256, h_ref, 256, bias=bd, act=1)
    ref = h_ref.double().cpu() @ Wo.double().T + bo.double()
    scale = (h_ref.double().cpu().abs() @ Wo.double().abs().T + bo.double().abs())
    for keep in (True, False):
        hid = torch.full((M, 256), float("nan"), device=DEV) if keep else None
        out = torch.full((M + 1, 6), -7.0, device=DEV)
        engine.last2(M, Ad, Wd, bd, Wod, bod, hid, out, 6, 1)
        if keep:
            assert torch.equal(hid, h_ref)
        got = out[:M, 1:1 + E].double().cpu()
        assert float(((got - ref).abs() / scale).max()) <= 2e-6, (M, E, keep)
        assert bool((out[M:] == -7.0).all()) and bool((out[:, 0] == -7.0).all()) and bool((out[:, 1 + E:] == -7.0).all())


@pytest.mark.parametrize("M", [1, 63, 65, 130, 5000, 70001])
def test_appearance_head_last_two_layers_fused(M):
    """clift_app_head_last2_fwd (128 -> 128 + ReLU, 128 -> 3, sigmoid in one launch) against clift_gemm x 2 + clift_rows_act_fwd: the
    hidden activation bit-identical, rgb to summation-order round-off (vs fp64), with and without writing the hidden activation."""
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(M * 3 + 1)
    H1 = torch.relu(torch.randn((M, 128), generator=g))
    W2 = (torch.randn((128, 128), generator=g) / 11).contiguous()
    b2 = torch.randn(128, generator=g) * 0.2
    W3 = (torch.randn((3, 128), generator=g) / 8).contiguous()
    b3 = torch.randn(3, generator=g) * 0.3
    H1d, W2d, b2d, W3d, b3d = (t.to(DEV) for t in (H1, W2, b2, W3, b3))
    H2_ref = torch.empty((M, 128), device=DEV)
    with engine.exact_fp32():            # (clift_app_head_last2_fwd is the exact-fp32 kernel: so is its reference; the default arithmetic has its own pair, tests/test_gpu_round5b.py)
        engine.gemm(M, 128, 128, H1d, 128, W2d, 128, H2_ref, 128, bias=b2d, act=1)
    pre = H2_ref.double().cpu() @ W3.double().T + b3.double()
    ref = torch.sigmoid(pre)
    for keep in (True, False):
        H2 = torch.full((M, 128), float("nan"), device=DEV) if keep else None
        rgb = torch.full((M + 1, 3), -7.0, device=DEV)
        engine.app_last2(M, H1d, W2d, b2d, W3d, b3d, H2, rgb)
        if keep:
            assert torch.equal(H2, H2_ref)
        rel_close(rgb[:M], ref, 1e-5, atol=2e-6, what=f"fused appearance output M={M}")
        assert bool((rgb[M:] == -7.0).all())


@pytest.mark.parametrize("M", [4096, 4129, 70001])
def test_unmasked_narrow_dgrad_stream(M):
    """k_dgrad_narrow_stream<.., MASK = false>: dF = dfeat Wb (K = 27 in a 28-float pitch -> N = 144), the dgrad of the appearance basis,
    against fp64; pad column of dfeat holds garbage-free zeros in the pipeline but is given non-zero values here (its weights are
    zero past K); rows beyond M and columns beyond N untouched; and N = 256, K = 22."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M)
    for N, K, lda in ((144, 27, 28), (256, 22, 24), (64, 3, 4)):
        A = torch.randn((M, lda), generator=g)
        W = (torch.randn((K, N), generator=g) / 4).contiguous()
        out = torch.full((M + 2, N + 4), -7.0, device=DEV)
        engine.gemm(M, N, K, A.to(DEV), lda, W.to(DEV), N, out, N + 4, b_trans=1)
        ref = A[:, :K].double() @ W.double()
        rel_close(out[:M, :N], ref, 2e-5, atol=2e-5 * float(ref.abs().max()), what=f"unmasked narrow dgrad N={N} K={K}")
        assert bool((out[M:] == -7.0).all()) and bool((out[:, N:] == -7.0).all())


@pytest.mark.parametrize("M", [4096, 4129, 70001])
def test_masked_narrow_dgrad_stream_128_wide(M):
    """k_dgrad_narrow_stream with N = 128: dH2 = mask(H2) . (dpre W3), the dgrad of the appearance output layer (K = 3 in a 4-float pitch)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + 1)
    A = torch.randn((M, 4), generator=g); A[:, 3] = 0
    W = (torch.randn((3, 128), generator=g) / 2).contiguous()
    mask = torch.randn((M, 128), generator=g)
    out = torch.full((M + 2, 132), -7.0, device=DEV)
    engine.gemm(M, 128, 3, A.to(DEV), 4, W.to(DEV), 128, out, 132, b_trans=1, mask=mask.to(DEV), ldmask=128)
    ref = (A[:, :3].double() @ W.double()) * (mask.double() > 0)
    rel_close(out[:M, :128], ref, 2e-5, atol=2e-5 * float(ref.abs().max()), what="masked narrow dgrad N=128")
    assert bool((out[M:] == -7.0).all()) and bool((out[:, 128:] == -7.0).all())


# ============================================================================ bf16 mode: a whole xyz head forward in one launch
@pytest.mark.parametrize("M", [1, 63, 65, 5000, 140001, 600000])
def test_bf16_fused_head_forward_matches_per_layer_path(M):
    """clift_xyz_head_bf16_fwd (K = 3 layer + two 256 x 256 bf16 layers (+ the E-wide output layer) with the activations resident in LDS)
    against the per-layer bf16 path (clift_linear_k3_fwd + k_layer_bf16 + narrow GEMM): same arithmetic => the kept activations are
    bit-identical, the output layer agrees to summation-order round-off; instance head (E = 3, everything fused) and semantic head
    (C = 22: three layers fused, fourth hidden layer and output layer per layer); with and without keeping the activations."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + 17)
    xa = torch.zeros((M, 4)); xa[:, :3] = torch.rand((M, 3), generator=g) * 2 - 1
    xa = xa.to(DEV)

    def lin(o, i, s):
        W = torch.zeros((o, (i + 3) // 4 * 4)); W[:, :i] = torch.randn((o, i), generator=g) * s
        return W.to(DEV)[:, :i], (torch.randn(o, generator=g) * 0.2).to(DEV)
    prev = engine.set_mlp_precision("bf16")
    try:
        for n_out, n_hidden in ((3, 3), (22, 4)):
            layers = [lin(256, 3, 0.8)] + [lin(256, 256, 0.09) for _ in range(n_hidden - 1)] + [lin(n_out, 256, 0.1)]
            res = {}
            for fused in (False, True):
                engine.FUSE_HEAD_BF16 = fused
                for keep in (True, False):
                    out = torch.full((M, n_out + 3), -7.0, device=DEV)
                    acts = engine.xyz_mlp_fwd(layers, xa, M, out, n_out + 3, 1, keep_first=keep)
                    res[(fused, keep)] = (out.clone(), acts)
            ref_out, ref_acts = res[(False, True)]
            for keep in (True, False):
                out, acts = res[(True, keep)]
                sc = float(ref_out[:, 1:1 + n_out].abs().max())
                assert float((out[:, 1:1 + n_out] - ref_out[:, 1:1 + n_out]).abs().max()) <= 2e-5 * sc + 1e-7, (M, n_out, keep)
                assert bool((out[:, 0] == -7.0).all()) and bool((out[:, 1 + n_out:] == -7.0).all())
                if keep:
                    assert len(acts) == len(ref_acts)
                    for a, b in zip(acts, ref_acts):
                        assert a.dtype == torch.bfloat16 and torch.equal(a, b)
                else:
                    assert acts == [None]
    finally:
        engine.FUSE_HEAD_BF16 = True
        engine.set_mlp_precision(prev)


# ============================================================================ sync-free steps (device-side row count)
def test_sync_free_training_steps_match_synchronising_ones():
    """config.nosync: after two learning steps the trainer never reads the active-sample count back -- buffers and grids are sized by a
    capacity, every per-sample kernel clamps to the count on the device (clift_bind_rows_limit) and the persistent kernels re-balance
    their row ranges over it.  Same batches, jitter and white-background flags through a default and a sync-free trainer: losses
    equal, gradients equal up to accumulation order, parameters after six steps within a fraction of a learning-rate step; chunked
    main pass (two chunks: the limit is handed back and forth between chunks for the backward); no overflow; the limit is INT_MAX again
    after every pass."""
    from contrastive_lift_amd import engine, synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    g = torch.Generator().manual_seed(5)
    jit = [torch.rand(2048, generator=g).to(DEV) for _ in range(6)]
    jit_i = [torch.rand(512, generator=g).to(DEV) for _ in range(6)]
    runs = {}
    for nosync in (False, True):
        for chunk in (0, 1024):
            model, renderer, pool = synthetic.make_scene(grid=64, num_classes=6, max_instances=3, seed=9, device=DEV, image=128, n_cams=2)
            cfg = default_config(chunk=chunk, instance_optimization_epoch=0, late_semantic_optimization=0, nosync=nosync)
            tr = HotPathTrainer(model, renderer, cfg, current_epoch=4)
            batches = [synthetic.make_batches(pool, 2048, 512, 6, 9, seed=40 + i, device=DEV) for i in range(3)]
            hist = []
            for step in range(6):
                b = batches[step % 3]
                tr.main_pass(b[0], jitter=jit[step], white_bg=bool(step % 2))
                gm = {k: v.detach().clone() for k, v in model.named_grad_views().items() if not k.startswith("render_instance")}
                lm = tr.losses.clone()
                tr.instance_pass(b[1], jitter=jit_i[step])
                hist.append((gm, lm, float(tr.losses[3])))
                if nosync:
                    assert int(engine.rows_limit(torch.device(DEV, torch.cuda.current_device()))[0]) == engine.INT_MAX
            if nosync:
                assert tr.overflow_steps == 0
                caps = {k: v["cap"] for k, v in tr._caps.items()}
                assert all(c is not None and c % 4096 == 0 for c in caps.values()), caps
            runs[(nosync, chunk)] = (hist, model.param_flat.detach().clone())
    for chunk in (0, 1024):
        ref, got = runs[(False, chunk)], runs[(True, chunk)]
        for step in (0, 3, 5):                      # 0: learning step (synchronising in both), 3 and 5: sync-free
            gm0, lm0, li0 = ref[0][step]
            gm1, lm1, li1 = got[0][step]
            rel_close(lm1[:3], lm0[:3], 2e-4, what=f"losses step {step} chunk {chunk}")
            rel_close(li1, li0, 2e-3, what=f"slow-fast loss step {step} chunk {chunk}")
            if step == 3:         # (the two runs' parameters already differ by accumulation-order round-off of three Adam steps: a few
                for k in gm0:     # ReLU-kink / activity-threshold flips are expected, hence the wider outlier allowance: seen 16 of 3072)
                    grad_close(gm1[k], gm0[k], what=f"sync-free grad {k} (chunk {chunk})", outlier_frac=2e-2, outlier_cap=1e-1)
        assert float((got[1] - ref[1]).abs().max()) <= 0.5 * 1e-2


def test_sync_free_pass_with_a_synchronising_segment_chunk_of_varying_size():
    """the segment term runs inside the main pass, and its capacity key (pass, n_rays) changes whenever the segment
    batch changes size -- such a chunk is marched WITHOUT a cap while the library-wide row limit still holds the previous capped chunk's
    count.  Before the fix its per-sample kernels were silently clamped to that count.  Segment batches of 700 / 1500 / 900 / 1300 rays
    behind capped main chunks: the segment loss and the semantic-head gradients must equal those of a synchronising trainer."""
    from contrastive_lift_amd import engine, synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    g = torch.Generator().manual_seed(11)
    sizes = [900, 1100, 700, 1500, 900, 1300]
    jit = [torch.rand(512, generator=g).to(DEV) for _ in sizes]           # a SMALL main batch: its capped row count is far below the segment's
    sjit = [torch.rand(n, generator=g).to(DEV) for n in sizes]
    runs = {}
    for nosync in (False, True):
        model, renderer, pool = synthetic.make_scene(grid=64, num_classes=6, max_instances=3, seed=9, device=DEV, image=128, n_cams=2)
        cfg = default_config(chunk=0, instance_optimization_epoch=0, late_semantic_optimization=0, nosync=nosync, segment_optimization_epoch=0)
        tr = HotPathTrainer(model, renderer, cfg, current_epoch=4)
        gsel = torch.Generator().manual_seed(3)
        hist = []
        for step, n in enumerate(sizes):
            b = synthetic.make_batches(pool, 512, 256, 6, 9, seed=60 + step, device=DEV)
            idx = torch.randint(0, pool.shape[0], (n,), generator=gsel).to(DEV)
            seg = dict(rays=pool[idx].contiguous(), group=torch.randint(0, 5, (n,), generator=gsel).to(DEV),
                       confidences=torch.rand(n, generator=gsel).to(DEV), n_groups=5)
            tr.main_pass(b[0], jitter=jit[step], white_bg=False, segments=seg, segment_jitter=sjit[step])
            gm = {k: v.detach().clone() for k, v in model.named_grad_views().items() if k.startswith("render_semantic")}
            hist.append((float(tr.loss_segment[0]), gm))
            if nosync:
                assert int(engine.rows_limit(torch.device(DEV, torch.cuda.current_device()))[0]) == engine.INT_MAX
        if nosync:
            assert tr.overflow_steps == 0
            assert any(v["cap"] is not None for k, v in tr._caps.items() if k[0] == "main"), "the main chunk never went sync-free"
        runs[nosync] = hist
    for step in range(len(sizes)):
        l0, g0 = runs[False][step]
        l1, g1 = runs[True][step]
        if step <= 3:      # later steps: the two runs' parameters differ by accumulation-order round-off of several Adam steps
            rel_close(l1, l0, 5e-4, what=f"segment loss step {step}")
        if step == 3:      # sync-free main chunk (capped at ~its own few thousand rows), 1500-ray segment chunk behind it
            for k in g0:
                grad_close(g1[k], g0[k], what=f"semantic-head grad {k} with a segment chunk behind a capped chunk", outlier_frac=2e-2, outlier_cap=1e-1)


# ============================================================================ appearance table gradients: float4-lane walk vs scalar-lane walk
@pytest.mark.parametrize("res,N", [((20, 28, 36), 700), ((64, 64, 64), 2048), ((128, 128, 128), 4096)])
def test_appearance_scatter_four_channel_lanes_match_one_channel_lanes(res, N, monkeypatch):
    """clift_app_gather_bwd with xa (a lane owns four channels, positions read from the forward's xa) against the same entry point without xa
    (one channel per lane, positions re-derived from the rays): the same per-sample terms merged along the same ray segments, so every
    appearance plane / line gradient agrees to fp32 summation order (1e-4 relative + 2e-5 of the tensor's scale).  Non-cubic grid (every
    plane with its own W x H and line length), a 64^3 grid and the bench's 128^3 / 4096 rays."""
    cl, op, orender, ofld, olosses, orays = _import()
    from contrastive_lift_amd import engine
    C_, E = 9, 3
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P, rays, rng = scene(op, orays, 77, res, C_, E, N, img=64, amp=3.0, sg=0.35)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    got = {}
    for with_xa in (False, True):
        monkeypatch.setattr(engine, "APP_SCATTER_XA", with_xa)
        m = build_model(cl, P, res, C_, E, -3.0, "softmax")
        r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
        _, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [1.0])
        got[with_xa] = {k: (None if g is None else g.detach().cpu()) for k, g in grads.items()}
    n_app = 0
    for k, ref in got[False].items():
        if k.startswith(("appearance_plane", "appearance_line")):
            assert float(ref.abs().max()) > 0, k
            grad_close(got[True][k], ref, what=f"float4-lane scatter {k}", rtol=1e-4, scale_atol=2e-5, outlier_frac=0.0, outlier_cap=2e-4)
            n_app += 1
    assert n_app == 6


@pytest.mark.parametrize("res,N", [((20, 28, 36), 700), ((64, 64, 64), 2048), ((128, 128, 128), 4096)])
def test_density_scatter_wave_walk_matches_group_walk(res, N, monkeypatch):
    """clift_density_bwd in its wave-per-(ray, 32-sample chunk) form (index work of the chunk done once, serial walk with per-channel work
    only) against the group-per-4-sample-segment walk (CLIFT_DENS_SCATTER=walk), with and without the forward's sigma: the same per-sample
    terms up to fp32 round-off, merged over longer runs, so every density plane / line gradient agrees to fp32 summation order."""
    cl, op, orender, ofld, olosses, orays = _import()
    C_, E = 9, 3
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P, rays, rng = scene(op, orays, 78, res, C_, E, N, img=64, amp=3.0, sg=0.35)
    jitter = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32))
    cots = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((N, 3), (N, C_), (N, 2 * E))]
    got = {}
    from contrastive_lift_amd import engine
    for mode in ("walk", "wave", "wave_nosigma"):
        monkeypatch.setenv("CLIFT_DENS_SCATTER", mode.split("_")[0])
        monkeypatch.setattr(engine, "DENS_BWD_SIGMA", mode != "wave_nosigma")
        m = build_model(cl, P, res, C_, E, -3.0, "softmax")
        r = cl.TensoRFRenderer(aabb, list(res), semantic_weight_mode="softmax").to(DEV)
        _, grads = _run_forward_backward(cl, m, r, rays, jitter, False, cots + [1.0])
        got[mode] = {k: (None if g is None else g.detach().cpu()) for k, g in grads.items()}
    n = 0
    for k, ref in got["walk"].items():
        if k.startswith(("density_plane", "density_line")):
            assert float(ref.abs().max()) > 0, k
            for mode in ("wave", "wave_nosigma"):
                grad_close(got[mode][k], ref, what=f"{mode} density scatter {k}", rtol=1e-4, scale_atol=2e-5, outlier_frac=0.0, outlier_cap=2e-4)
            n += 1
    assert n == 6


@pytest.mark.parametrize("M,no,ldd,ni", [(4099, 3, 4, 128), (249003, 3, 4, 128), (40001, 27, 28, 144), (8191, 27, 28, 144), (5000, 22, 24, 36), (4096, 8, 8, 252)])
def test_narrow_wgrad_stream_on_narrower_activations(M, no, ldd, ni):
    """k_wgrad_narrow_stream with X narrower than 256 columns (the appearance output layer: 3 x 128; the appearance basis matrix: 27 x 144):
    waves whose 32 columns start past the width only copy and wait.  Against fp64 and against the VALU kernel (CLIFT_NO_PERSISTENT),
    accumulating onto existing contents, with and without a bias gradient, ragged row counts."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + no + ni)
    dY = torch.zeros((M, ldd))
    dY[:, :no] = torch.randn((M, no), generator=g)
    X = torch.relu(torch.randn((M, ni), generator=g)).to(DEV)
    dYd = dY.to(DEV)
    refw = dY[:, :no].double().T @ X.cpu().double() + 0.25
    refb = dY[:, :no].double().sum(0) - 1.0
    outs = []
    for valu in (False, True):
        if valu:
            os.environ["CLIFT_NO_PERSISTENT"] = "1"
        try:
            gW = torch.full((no, ni), 0.25, device=DEV)
            gb = torch.full((no,), -1.0, device=DEV)
            engine.wgrad(no, ni, M, dYd, ldd, X, ni, gW, gb)
            gW_nb = torch.full((no, ni), 0.25, device=DEV)
            engine.call("clift_wgrad_narrow", engine.ptr(dYd), ldd, no, engine.ptr(X), ni, ni, M, engine.ptr(gW_nb), ni, None, 0, engine.stream())
        finally:
            os.environ.pop("CLIFT_NO_PERSISTENT", None)
        rel_close(gW, refw, 2e-5, atol=2e-5 * float(refw.abs().max()), what=f"narrow wgrad ni={ni} valu={valu}")
        rel_close(gb, refb, 2e-5, atol=2e-5 * M ** 0.5, what="bias sums")
        rel_close(gW_nb, refw, 2e-5, atol=2e-5 * float(refw.abs().max()), what="no-bias form")
        outs.append(gW)
    rel_close(outs[0], outs[1], 2e-5, atol=2e-5 * float(refw.abs().max()), what="stream vs VALU kernel")


@pytest.mark.parametrize("M", [4096, 4099, 40001, 249003])
def test_first_appearance_layer_input_gradient_split_launch(M):
    """dX = dH (M x 128) W (128 x 160), unmasked (the first appearance layer's input gradient, tensoRF.py:389-393 backward): columns 0..127 run
    through the persistent 128-wide dgrad kernel without a mask, columns 128..159 through the tiled 32-column kernel.  Against fp64; pad
    columns beyond the pitch are never written."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M)
    dH = torch.randn((M, 128), generator=g)
    W = torch.randn((128, 160), generator=g) * 0.1
    ref = dH.double() @ W.double()
    out = torch.full((M, 164), -7.0, device=DEV)
    engine.gemm(M, 160, 128, dH.to(DEV), 128, W.to(DEV), 160, out, 164, b_trans=1)
    rel_close(out[:, :160], ref, 2e-5, atol=2e-5 * float(ref.abs().max()), what="split dX")
    assert bool((out[:, 160:] == -7.0).all())


@pytest.mark.parametrize("M,no,ldd", [(4096, 22, 24), (4099, 3, 4), (40001, 22, 24), (249003, 3, 4), (5000, 32, 32), (8191, 6, 8), (33, 22, 24)])
def test_output_layer_backward_in_one_pass(M, no, ldd):
    """clift_out_layer_bwd: dX = (H > 0) . (dOut W), gW += dOut^T H, gb += column sums of dOut from ONE pass over the hidden activation H.
    Against fp64 (accumulating onto existing gW / gb, ragged row counts incl. a last partial tile, zero pad columns of dOut), and dX
    bit-identical to the separate masked dgrad launch (the same MFMA chain)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(M + no)
    dOut = torch.zeros((M, ldd))
    dOut[:, :no] = torch.randn((M, no), generator=g)
    W = torch.randn((no, 256), generator=g) * 0.1
    H = torch.relu(torch.randn((M, 256), generator=g))
    dOd, Wd, Hd = dOut.to(DEV), W.to(DEV), H.to(DEV)
    dX = torch.full((M, 256), -7.0, device=DEV)
    gW = torch.full((no, 256), 0.25, device=DEV)
    gb = torch.full((no,), -1.0, device=DEV)
    engine.call("clift_out_layer_bwd", engine.ptr(dOd), ldd, no, engine.ptr(Wd), 256, engine.ptr(Hd), 256, M, engine.ptr(dX), 256,
                engine.ptr(gW), 256, engine.ptr(gb), engine.stream())
    ref_dx = (dOut[:, :no].double() @ W.double()) * (H > 0)
    ref_w = dOut[:, :no].double().T @ H.double() + 0.25
    ref_b = dOut[:, :no].double().sum(0) - 1.0
    rel_close(dX, ref_dx, 2e-5, atol=2e-5 * float(ref_dx.abs().max()), what="fused dX")
    rel_close(gW, ref_w, 2e-5, atol=2e-5 * float(ref_w.abs().max()), what="fused gW")
    rel_close(gb, ref_b, 2e-5, atol=2e-5 * M ** 0.5, what="fused gb")
    if M >= 4096:
        dX2 = torch.empty((M, 256), device=DEV)
        engine.gemm(M, 256, no, dOd, ldd, Wd, 256, dX2, 256, b_trans=1, mask=Hd, ldmask=256)
        assert torch.equal(dX, dX2)
