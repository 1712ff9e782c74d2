"""Determinism soak: is the hot path bit-reproducible in ONE process on ONE device?

    python tools/determinism_soak.py <tag> <mode> <n_renders> <n_steps> [rays_per_render=131072] [chunk=65536]

Render phase: the same rays of the 128^3 bench scene (inference form: step ratio halved, grad_heads=(), the configs[4] chunk shape) are
rendered n_renders times; EVERY tensor of every chunk -- the four outputs and the chain's intermediates (sigma, weights, compacted list,
positions, the appearance front end's feat / X rows, the first appearance hidden layer, the three heads' per-sample outputs) -- is compared
bit for bit with the first render's.  The intermediates are references the engine keeps when ``engine.SOAK_KEEP`` is set: same launches,
same kernels, nothing re-run.  A differing tensor is reported with its name, the number of differing elements / rows and the first index,
so that the FIRST stage of a chain that went wrong names the kernel.

Step phase: one training step (4096 + 1024 rays, the bench batch) is replayed n_steps times from one snapshot of parameters and optimizer
state, with fixed jitter and background; compared bit for bit: every forward activation the contexts retain and every backward temporary
(pre-activation gradients, per-layer input gradients, d sigma).  Sums that go through floating-point atomics (weight / table gradients) are
order-dependent by construction: for those the largest deviation from the first replay, relative to the largest entry, is reported.

Two instances at once (tags A and B) is the device-sharing experiment of rounds 3 / 4; one instance alone is the control this tool is for."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastive_lift_amd import engine, synthetic
from contrastive_lift_amd.trainer import HotPathTrainer, default_config

tag, mode, n_renders, n_steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
n_rays = int(sys.argv[5]) if len(sys.argv) > 5 else 131072
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 65536
dev = torch.device("cuda:0")
engine.set_mlp_precision(mode)
engine.SOAK_KEEP = True
P = lambda *a: print(tag, mode, *a, flush=True)


def describe(a, b):
    d = (a != b)
    if a.dtype.is_floating_point:          # NaN != NaN: compare the bits
        d = a.view(torch.int32 if a.element_size() == 4 else torch.int16) != b.view(torch.int32 if b.element_size() == 4 else torch.int16)
    n = int(d.sum())
    if n == 0:
        return None
    idx = torch.nonzero(d)
    rows = int(torch.unique(idx[:, 0]).numel())
    first = idx[0].tolist()
    fa, fb = a[tuple(first)], b[tuple(first)]
    return f"ndiff {n} rows {rows} of {tuple(a.shape)} first {first} got {float(fa)!r} want {float(fb)!r}"


def compare(it, got, ref, bad, phase):
    first_bad = None
    for name in ref:
        a, b = got.get(name), ref[name]
        if a is None or a.shape != b.shape:
            bad[name] = bad.get(name, 0) + 1
            P(phase, "iter", it, name, "MISSING / shape", None if a is None else tuple(a.shape), tuple(b.shape))
            continue
        if torch.equal(a, b):
            continue
        msg = describe(a, b)
        if msg is None:
            continue
        bad[name] = bad.get(name, 0) + 1
        if first_bad is None:
            first_bad = name
        if bad[name] <= 6:
            P(phase, "iter", it, name, msg)
    return first_bad


# ------------------------------------------------------------------------------------------------ render phase
def ctx_tensors(o, ctx, prefix):
    t = {f"{prefix}out.rgb": o["rgb"], f"{prefix}out.sem": o["semantics"], f"{prefix}out.inst": o["instances"], f"{prefix}out.depth": o["depth"],
         f"{prefix}out.opacity": o["opacity"]}
    for k in ("sigma", "w", "act_idx", "ray_start", "xa", "rgb_s", "sem_s", "inst_s", "F", "feat", "X", "H1", "H2"):
        v = getattr(ctx, k, None)
        if torch.is_tensor(v):
            t[prefix + k] = v
    for k, v in (getattr(ctx, "soak", None) or {}).items():
        if torch.is_tensor(v):
            t[prefix + "app." + k] = v
    for head in ("sem_acts", "inst_fast_acts", "inst_slow_acts"):
        for i, v in enumerate(getattr(ctx, head, None) or []):
            if torch.is_tensor(v):
                t[f"{prefix}{head}[{i}]"] = v
    return t


model, renderer, pool = synthetic.make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device=dev)
train_ratio = renderer.step_ratio
if n_renders > 0:
    renderer.update_step_ratio(train_ratio * 0.5)
    rays = pool[:n_rays].contiguous()

    def render():
        t = {}
        for ci, i in enumerate(range(0, rays.shape[0], chunk)):
            o, ctx = engine.render_forward(model, renderer, rays[i:i + chunk], None, False, grad_heads=())
            t.update(ctx_tensors(o, ctx, f"c{ci}."))
        return t

    ref = {k: v.clone() for k, v in render().items()}
    torch.cuda.synchronize()
    P("render phase:", len(ref), "tensors per render,", sum(v.numel() * v.element_size() for v in ref.values()) >> 20, "MiB compared per render; M per chunk",
      [int(v.shape[0]) for k, v in ref.items() if k.endswith(".xa")])
    bad, firsts, t0 = {}, {}, time.time()
    for it in range(n_renders):
        fb = compare(it, render(), ref, bad, "render")
        if fb is not None:
            firsts[fb] = firsts.get(fb, 0) + 1
        if (it + 1) % 1000 == 0:
            P(f"render {it + 1}/{n_renders} {time.time() - t0:.0f}s bad so far:", bad)
    P(f"RENDER RESULT {n_renders} renders of {n_rays} rays (chunk {chunk}) in {time.time() - t0:.1f}s: renders with a differing tensor, by tensor:", bad or "none",
      "| first differing tensor of a bad render:", firsts or "none")
    del ref
    renderer.update_step_ratio(train_ratio)
    torch.cuda.empty_cache()

# ------------------------------------------------------------------------------------------------ training-step phase
if n_steps > 0:
    cfg = default_config(chunk=4096, batch_size=4096, max_rays_instances=1024, mlp_dtype=mode, late_semantic_optimization=0, instance_optimization_epoch=0)
    tr = HotPathTrainer(model, renderer, cfg, current_epoch=4)
    S = int(renderer.n_samples)
    batches = synthetic.make_batches(pool, 4096, 1024, 22, 40, seed=1, device=dev)
    g = torch.Generator(device="cpu").manual_seed(7)
    jit_main = torch.rand(4096, generator=g).to(dev)
    jit_inst = torch.rand(1024, generator=g).to(dev)
    for _ in range(3):                      # a few real steps first: Adam moments and the slow net are not at their initial values
        tr.main_pass(batches[0], jitter=jit_main, white_bg=False)
        tr.instance_pass(batches[1], jitter=jit_inst)
    snap = dict(p=model.param_flat.detach().clone(), m0=tr.opt_main.m.clone(), v0=tr.opt_main.v.clone(), t0=dict(tr.opt_main.t),
                m1=tr.opt_inst.m.clone(), v1=tr.opt_inst.v.clone(), t1=dict(tr.opt_inst.t))

    captured = []
    orig_join = engine.Branches.join

    def join(self):
        captured.append(list(self.keep))
        orig_join(self)
    engine.Branches.join = join
    orig_dens = engine._density_backward

    def dens(model_, ctx, views, gviews, g_w, g_op, g_dist, keep):
        orig_dens(model_, ctx, views, gviews, g_w, g_op, g_dist, keep)
        captured.append([g_w, g_op] + list(keep[-1:]))
    engine._density_backward = dens
    fwd_ctx = []
    orig_rf, orig_ff = engine.render_forward, engine.feature_forward

    def rf(*a, **k):
        o, ctx = orig_rf(*a, **k)
        fwd_ctx.append((o, ctx))
        return o, ctx

    def ff(*a, **k):
        o, ctx = orig_ff(*a, **k)
        oo = dict(rgb=None, semantics=None, instances=o[0] if isinstance(o, tuple) else None, depth=ctx.ray_out[:, 1], opacity=ctx.ray_out[:, 0])
        fwd_ctx.append((oo, ctx))
        return o, ctx
    engine.render_forward, engine.feature_forward = rf, ff

    def restore():
        with torch.no_grad():
            model.param_flat.copy_(snap["p"])
        tr.opt_main.m.copy_(snap["m0"]); tr.opt_main.v.copy_(snap["v0"]); tr.opt_main.t = dict(snap["t0"])
        tr.opt_inst.m.copy_(snap["m1"]); tr.opt_inst.v.copy_(snap["v1"]); tr.opt_inst.t = dict(snap["t1"])

    def step():
        # Both passes start from the SAME snapshot: the main pass ends with an Adam step on gradients that were summed through floating-point
        # atomics (order-dependent in the last bits), so an instance pass run behind it sees slightly different tables in every replay -- the
        # first version of this tool did that and every instance-pass tensor "differed" (profiles/r05_determinism_before_fix.txt)
        restore()
        del captured[:], fwd_ctx[:]
        t, soft = {}, {}
        tr.main_pass(batches[0], jitter=jit_main, white_bg=False)
        soft["grad.main"] = model.grad_flat[tr.main_range[0]:tr.main_range[1]].clone()
        soft["param.after_main"] = model.param_flat.detach().clone()
        restore()
        tr.instance_pass(batches[1], jitter=jit_inst)
        soft["grad.inst"] = model.grad_flat[tr.inst_range[0]:tr.inst_range[1]].clone()
        soft["param.after_inst"] = model.param_flat.detach().clone()
        for pi, (o, ctx) in enumerate(fwd_ctx):
            for k, v in ctx_tensors({k_: o.get(k_) for k_ in ("rgb", "semantics", "instances", "depth", "opacity")}, ctx, f"p{pi}.").items():
                if v is not None:
                    t[k] = v
        for bi, keep in enumerate(captured):
            for ki, v in enumerate(keep):
                if torch.is_tensor(v):
                    t[f"bwd{bi}.{ki}{tuple(v.shape)}"] = v
        soft["loss values"] = tr.losses.clone()      # (block partial sums folded with floating-point atomics: order-dependent in the last bits, like the gradients)
        return t, soft

    ref, soft_ref = step()
    ref = {k: v.clone() for k, v in ref.items()}
    torch.cuda.synchronize()
    P("step phase:", len(ref), "bit-compared tensors per step,", sum(v.numel() * v.element_size() for v in ref.values()) >> 20, "MiB; M of the passes",
      [int(v.shape[0]) for k, v in ref.items() if k.endswith(".xa")])
    bad, firsts, soft_max, t0 = {}, {}, {k: 0.0 for k in soft_ref}, time.time()
    scale = {k: float(v.abs().max()) for k, v in soft_ref.items()}
    for it in range(n_steps):
        got, soft = step()
        fb = compare(it, got, ref, bad, "step")
        if fb is not None:
            firsts[fb] = firsts.get(fb, 0) + 1
        for k in soft_ref:
            soft_max[k] = max(soft_max[k], float((soft[k] - soft_ref[k]).abs().max()) / max(scale[k], 1e-30))
        if (it + 1) % 500 == 0:
            P(f"step {it + 1}/{n_steps} {time.time() - t0:.0f}s bad so far:", bad)
    P(f"STEP RESULT {n_steps} replays of one training step (4096 + 1024 rays) in {time.time() - t0:.1f}s: steps with a differing deterministic tensor, by tensor:",
      bad or "none", "| first differing tensor of a bad step:", firsts or "none",
      "| atomically accumulated sums, largest |delta| / max|entry| over all replays:", {k: f"{v:.2e}" for k, v in soft_max.items()})
