"""Training step of the hot path -- mirror of reference trainer/train_panopli_tensorf.py:98-103 (optimizers),
:105-146 (chunked forwards), :148-228 (training_step), :256-310 + :325-329 (slow-fast loss + EMA), :446-447
(dist-reg ramp), driven directly through engine.py (no autograd graph, no Lightning).

Data-parallel: one process per GPU; every rank renders its own shard of the ray batch; after each backward ONE
all-reduce (RCCL over xGMI, `torch.distributed` backend "nccl") sums the contiguous gradient range of the arena
that the pass touched -- the reference's DDP does the same exchange bucket by bucket, twice per step
(SURVEY 2a/2b).  The slow net is an EMA of synchronised fast weights, so it needs no exchange.
"""
import math
import os
import time
import types

import torch
import torch.distributed as dist

from . import _lib, engine
from .loss import contrastive_loss, linear_assignment_loss, slow_fast_loss


def default_config(**over):
    """Hot-path keys of config/template/panopli_paper.yaml with the contrastive-lift overlays
    (config/experiment/contrastive_lift*.yaml)."""
    c = dict(lr=5e-4, weight_decay=1e-8, lambda_rgb=1.0, lambda_semantics=0.1, lambda_dist_reg=0.005,
             lambda_tv_density=0.1, lambda_tv_appearance=0.01, lambda_tv_semantics=0.02, lambda_tv_instances=0.02,
             late_semantic_optimization=1, instance_optimization_epoch=3, chunk=2048, perturb=1.0, batch_size=2048,
             max_rays_instances=1024, max_instances=3, instance_loss_mode="slow_fast", use_DINO_style=True,
             semantic_weight_mode="softmax", stop_semantic_grad=True, probabilistic_ce_mode="TTAConf", weight_class_0=0.0,
             decay_step=[9, 10], decay_gamma=0.5, temperature=100.0,
             lambda_segment=1.2, segment_grouping_mode="argmax_conf", segment_optimization_epoch=6, batch_size_segments=32,
             max_rays_segments=1024, use_symmetric_ce=False, ce_alpha=0.85, ce_beta=0.15, reweight_fg=False,
             mlp_dtype="fp32x6",   # this build's extension key: arithmetic of the 256-wide MLP layers -- "fp32x6" (default: fp32-faithful 3 x bf16 split,
                                   # six products, fp32 accumulate), "fp32" (exact fp32 MFMA), "bf16" (bf16 operands, fp32 accumulate; BASELINE configs[2])
             nosync=False,         # this build's extension key: sync-free steps (no read-back of the active-sample count; exact-fp32 path)
             skip_discarded_instance_heads=False,    # extension key: do not evaluate the instance heads in the main pass, where the reference
                                                     # computes and discards them (T:155) -- same results, ~12 % less work; the train CLI sets it
             host_rng=False,       # extension key: draw the per-ray jitter from torch's CPU generator like the reference's renderer (R:808-810) instead of the
                                   # device generator; the train CLI sets it (the drop-in's random sources are the reference's), the benchmark does not
             grad_shards=True)     # extension key: MLP gradients accumulate in eight per-XCD copies, folded once per pass (include/clift.h, ABI 12)
    c.update(over)
    return types.SimpleNamespace(**c)


class ArenaAdam:
    """torch.optim.Adam semantics (trainer/__init__.py:134-135) over contiguous arena ranges: one clift_adam launch
    per (lr, weight_decay) range.  ``ranges`` = [(name, start, end, lr)].

    torch's Adam skips a parameter whose ``.grad`` is None (the reference zeroes with ``set_to_none=True``, T:152,211) and keeps
    a step count PER PARAMETER, so a head that has no gradient source yet (the semantic MLP while epoch <
    late_semantic_optimization) is neither decayed nor moved and starts its bias correction at t = 1 when its loss term
    switches on.  Here every range has its own step count and ``step(skip=...)`` leaves the named ranges untouched."""

    def __init__(self, model, ranges, betas, weight_decay, eps=1e-8):
        self.model, self.betas, self.wd, self.eps = model, betas, weight_decay, eps
        self.ranges = [tuple(r) for r in ranges]
        self.m = torch.zeros_like(model.param_flat)
        self.v = torch.zeros_like(model.param_flat)
        self.t = {r[0]: 0 for r in self.ranges}
        self.lr_scale = 1.0
        self.frozen = set()       # ranges whose parameters this optimizer no longer holds (reference: tables replaced by ``shrink`` with no
                                  # optimizer rebuild in the same hook, golden G21 scenario B) -- never stepped until the next rebuild

    def carry_from(self, old, old_arena):
        """After the arena was re-packed WITHOUT an optimizer rebuild (a bbox shrink in an epoch that has no grid upsample, T:448-449):
        ranges whose size did not change (the MLPs) keep their moments and step counts at their new offsets -- the reference's Adam
        still holds those very Parameter objects; ranges whose size changed (the cropped tables) were replaced by new Parameter
        objects the reference's optimizer never sees: frozen until the next ``setup_optimizers``."""
        prev = {r[0]: r for r in old.ranges}
        for name, a, b, lr in self.ranges:
            if name in prev and prev[name][2] - prev[name][1] == b - a and name not in old.frozen:
                pa, pb = prev[name][1], prev[name][2]
                self.m[a:b].copy_(old.m[pa:pb]); self.v[a:b].copy_(old.v[pa:pb])
                self.t[name] = old.t[name]
            else:
                self.frozen.add(name)
        self.lr_scale = old.lr_scale
        return self

    def step(self, skip=()):
        p, g = self.model.param_flat, self.model.grad_flat
        st = _lib.stream()
        for name, a, b, lr in self.ranges:
            if name in skip or name in self.frozen or b <= a:
                continue
            self.t[name] += 1
            _lib.call("clift_adam", _lib.ptr(p[a:b]), _lib.ptr(g[a:b]), _lib.ptr(self.m[a:b]), _lib.ptr(self.v[a:b]), b - a,
                      float(lr * self.lr_scale), self.betas[0], self.betas[1], self.eps, float(self.wd), self.t[name], st)

    def state_dict(self):
        return {"m": self.m.detach().clone(), "v": self.v.detach().clone(), "t": dict(self.t), "lr_scale": self.lr_scale, "frozen": sorted(self.frozen)}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"].to(self.m.device)); self.v.copy_(sd["v"].to(self.v.device))
        self.t = {k: int(sd["t"].get(k, 0)) for k in self.t}
        self.lr_scale = float(sd.get("lr_scale", 1.0))
        self.frozen = set(sd.get("frozen", ()))

    # ---- torch.optim.Adam's own state_dict layout (what a Lightning checkpoint of the reference holds under ``optimizer_states``)
    def _range_of_param(self, name):
        g = self.model.arena.by_name[name].group
        for r in self.ranges:
            if r[0] == g or (r[0] == "grids" and g in ("grid_density", "grid_app")):
                return r
        return None

    def torch_state_dict(self, groups):
        """``groups`` = [(lr, [parameter names])] in the order of the reference's param groups (tensoRF.py:199-246).  Parameters are
        numbered across groups; a parameter that was never stepped has no state entry, like torch's lazily created state."""
        mv, vv = self.model.arena.views(self.m), self.model.arena.views(self.v)
        state, pgs, k = {}, [], 0
        for lr, names in groups:
            ids = []
            for n in names:
                r = self._range_of_param(n)
                if r is not None and self.t[r[0]] > 0:
                    state[k] = {"step": torch.tensor(float(self.t[r[0]])), "exp_avg": mv[n].detach().cpu().contiguous().clone(),
                                "exp_avg_sq": vv[n].detach().cpu().contiguous().clone()}
                ids.append(k)
                k += 1
            pgs.append({"lr": float(lr * self.lr_scale), "initial_lr": float(lr), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": float(self.wd),
                        "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                        "params": ids})
        return {"state": state, "param_groups": pgs}

    def load_torch_state_dict(self, sd, groups):
        """Inverse of ``torch_state_dict``: moments of every parameter that has state; a range's step count = that of its parameters
        (they step together in the reference as well: one optimizer.step() per pass)."""
        mv, vv = self.model.arena.views(self.m), self.model.arena.views(self.v)
        names = [n for _, ns in groups for n in ns]
        if len(names) != sum(len(g["params"]) for g in sd["param_groups"]):
            raise ValueError(f"optimizer state has {sum(len(g['params']) for g in sd['param_groups'])} parameters, this field has {len(names)}")
        self.m.zero_(); self.v.zero_()
        self.t = {k: 0 for k in self.t}
        for k, n in enumerate(names):
            st = sd["state"].get(k)
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(mv[n].shape):
                raise ValueError(f"optimizer state of {n}: shape {tuple(st['exp_avg'].shape)} != {tuple(mv[n].shape)}")
            mv[n].copy_(st["exp_avg"].to(self.m.device)); vv[n].copy_(st["exp_avg_sq"].to(self.v.device))
            r = self._range_of_param(n)
            if r is not None:
                self.t[r[0]] = max(self.t[r[0]], int(float(st["step"])))


def _shard_guard(fn):
    def guarded(self, *a, **k):
        try:
            return fn(self, *a, **k)
        except BaseException:
            self._shards_off()
            raise
    guarded.__doc__, guarded.__name__ = fn.__doc__, fn.__name__
    return guarded


class HotPathTrainer:
    """Owns field + renderer + the two Adam optimizers; ``training_step`` = main pass + instance pass."""

    def __init__(self, model, renderer, config, class_weights=None, current_epoch=0, white_bg=False):
        self.model, self.renderer, self.config = model, renderer, config
        # config variants of the reference that this trainer does not implement fail loudly instead of being ignored
        unsupported = [(k, v) for k, v in (("use_distilled_features_semantic", False), ("use_distilled_features_instance", False),
                                            ("use_proj", False), ("use_feature_regularization", False))
                       if getattr(config, k, v) != v]
        if unsupported:
            raise NotImplementedError("HotPathTrainer: config options outside the contrastive-lift hot path: " +
                                      ", ".join(f"{k}={getattr(config, k)!r} (only {v!r} is built)" for k, v in unsupported))
        self.white_bg = bool(white_bg)            # dataset attribute in the reference (train_set.white_bg, T:109)
        engine.set_mlp_precision(getattr(config, "mlp_dtype", None) or engine.DEFAULT_MLP_DTYPE)    # process-wide switch of the matrix-core launches
        self.device = model.param_flat.device
        self.current_epoch = current_epoch
        C = model.num_semantic_classes
        cw = torch.ones(C) if class_weights is None else torch.as_tensor(class_weights, dtype=torch.float32).clone()
        if class_weights is None:
            cw[0] = config.weight_class_0                              # T:70
        self.class_weights = cw.to(self.device)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.setup_optimizers()
        self.on_train_epoch_start(maintenance=False)      # (the ramp only: the field handed in is at this epoch's resolution already)
        self.losses = torch.zeros(4, dtype=torch.float32, device=self.device)   # rgb, sem, tv, clustering (last step)
        # sync-free mode: per pass a capacity for the compacted buffers, learnt from the first (synchronising) steps and followed
        # asynchronously afterwards; see _capacity / _follow
        self.nosync = bool(getattr(config, "nosync", False)) and getattr(config, "mlp_dtype", None) in ("fp32", "f32", "fp32x6", None)
        self._caps = {}
        self.overflow_steps = 0

    # ------------------------------------------------------------------ optimizers (T:98-103)
    def setup_optimizers(self, carry=False):
        """T:98-103 (``configure_optimizers``): fresh Adam moments, step counts AND MultiStepLR schedulers -- Lightning's
        ``strategy.setup_optimizers`` (T:457,469) replaces both.  ``carry``: the arena was re-packed by a bbox shrink in an epoch without a
        grid upsample; the reference does NOT rebuild there (ArenaAdam.carry_from)."""
        m, c = self.model, self.config
        old = (getattr(self, "opt_main", None), getattr(self, "opt_inst", None), getattr(self, "_opt_arena", None)) if carry else None
        have = lambda *gs: [g for g in gs if g in m.arena.groups]                # (grid_sem / grid_inst exist only when that head sits on its own VM grid)
        a0, a1 = m.arena.range_of("grid_density", "grid_app")
        b0, b1 = m.arena.range_of("net_app")
        s0, s1 = m.arena.range_of("net_sem")
        main = [("grids", a0, a1, c.lr * 20), ("net_app", b0, b1, c.lr)]
        if have("grid_sem"):                                                     # tensoRF.py:205-208: the semantic tables at the grid rate
            main.append(("grid_sem",) + m.arena.range_of("grid_sem") + (c.lr * 20,))
        self.opt_main = ArenaAdam(m, main + [("net_sem", s0, s1, c.lr)], (0.9, 0.99), c.weight_decay)
        self.main_range = m.arena.range_of(*have("grid_density", "grid_app", "net_app", "grid_sem", "net_sem"))
        self.late_range = m.arena.range_of("grid_density")                       # final only after the density backward
        self.early_range = m.arena.range_of(*have("grid_app", "net_app", "grid_sem", "net_sem"))     # final once the head chains are issued
        # data-parallel exchange of the main pass: True = early range all-reduced asynchronously under the density backward, False = one
        # synchronous all-reduce after the backward, "auto" (default) = measure both over the first steps and keep the faster one.  The
        # persistent kernels hold one block per CU for a whole launch, so while the asynchronous collective is in flight they leave
        # ``allreduce_cu_reserve`` CUs to RCCL's kernels (clift_set_cu_reserve).
        ov = getattr(c, "overlap_allreduce", "auto")
        self.overlap_allreduce = ov if isinstance(ov, bool) else ("auto" if str(ov).lower() == "auto" else str(ov).lower() in ("1", "true", "yes"))
        self.allreduce_cu_reserve = int(getattr(c, "allreduce_cu_reserve", 8))
        self._cal = {"times": {True: [], False: []}, "decided": None}
        self.force_collectives = False          # tests: issue the collectives in a one-rank group too (RCCL path on a single-GPU box)
        # The slow MLP is listed in the reference's instance optimizer when not DINO-style (F:241-244), but its output is detached
        # in every loss mode (T:268), so its .grad stays None and torch's Adam never touches it: only the fast range is stepped.
        i0, i1 = m.arena.range_of("inst_fast")
        inst = [("inst_fast", i0, i1, c.lr)]
        if have("grid_inst"):                                                    # tensoRF.py:232-236: the instance tables at the grid rate, the basis matrix at the net rate
            inst = [("grid_inst",) + m.arena.range_of("grid_inst") + (c.lr * 20,), ("inst_basis",) + m.arena.range_of("inst_basis") + (c.lr,)] + inst
        self.opt_inst = ArenaAdam(m, inst, (0.9, 0.999), c.weight_decay)
        if old is not None and old[0] is not None:
            self.opt_main.wd = old[0].wd                   # (the optimizer objects survive: so does the weight decay they were built with)
            self.opt_inst.wd = old[1].wd
            self.opt_main.carry_from(old[0], old[2])
            self.opt_inst.carry_from(old[1], old[2])
        else:
            self.sched_steps = 0                           # MultiStepLR.last_epoch of the schedulers created with the optimizers
            self._apply_lr_schedule()
        self._opt_arena = m.arena
        self.inst_range = m.arena.range_of(*have("grid_inst", "inst_basis", "inst_fast"))
        # sync-free capacities were learnt for the previous grid / step size / bounding box: forget them (two synchronising steps again)
        if hasattr(self, "_caps"):
            self._caps = {}
        self._setup_grad_shards()

    # ------------------------------------------------------------------ XCD-private shards of the MLP gradients (include/clift.h, ABI 12)
    def _setup_grad_shards(self):
        """Eight copies (one per XCD) of the gradient range of the three trained MLPs (appearance, semantic, fast instance: ~1.5 MB): the
        weight-gradient kernels of a pass add into the copy of their block's XCD and one fold per pass adds the copies into the gradients --
        instead of 256 blocks of eight XCDs read-modify-writing the same cache lines at the end of every such launch (8 - 20 us each).
        ``grad_shards: False`` in the config turns it off."""
        m = self.model
        self._shards = None
        if self.device.type != "cuda" or not bool(getattr(self.config, "grad_shards", True)):
            return
        if "grid_sem" in m.arena.groups or "grid_inst" in m.arena.groups:
            return           # (a head on its own VM grid puts tables between the MLP ranges: the shard range would span them; those configs add directly)
        s0, s1 = m.arena.range_of("net_app", "net_sem", "inst_fast")
        n = s1 - s0
        stride = (n + 63) // 64 * 64                                   # floats; shards start on 256-byte boundaries
        self._shards = torch.zeros(8 * stride, dtype=torch.float32, device=self.device)
        self._shard_range = (s0, s1)
        g = m.grad_flat
        self._shard_src = torch.tensor([g.data_ptr() + 4 * s0, 4 * n, self._shards.data_ptr(), 4 * stride, 1], dtype=torch.int64, device=self.device)
        self._shard_grad_ptr = g.data_ptr()
        self._shard_record = engine.grad_shard_record(self.device)

    def _pass_begin(self, rng):
        """Clear the pass's gradient range; with shards: one launch that also switches them on for this pass's kernels."""
        m = self.model
        g = m.grad_flat[rng[0]:rng[1]]
        if self._shards is None:
            g.zero_()
            return
        if m.grad_flat.data_ptr() != self._shard_grad_ptr:             # the arena moved (grid resize without a new optimizer): re-describe it
            self._setup_grad_shards()
        if g.data_ptr() % 16 != 0:                                      # (the arena pads every tensor to 16 bytes; belt and braces)
            g.zero_()
            g = g[:0]
        _lib.call("clift_grad_shards_begin", _lib.ptr(self._shard_record), _lib.ptr(self._shard_src), _lib.ptr(g) if g.numel() else None, g.numel(),
                  _lib.stream())

    def _shards_off(self):
        """A pass that raised between begin and fold must not leave the shards switched on for whoever runs a backward next, nor the partial
        sums its kernels already flushed into the eight copies (the next pass's fold would add them to that step's gradients), nor the CUs
        reserved for an all-reduce that will not be waited for.  Runs inside an ``except``: whatever fails here (the device may be the
        reason we are here) must not replace the original exception."""
        try:
            if getattr(self, "allreduce_cu_reserve", 0) > 0 and self.device.type == "cuda":
                _lib.call("clift_set_cu_reserve", 0)
        except Exception:
            pass
        try:
            if getattr(self, "_shards", None) is not None:
                _lib.call("clift_grad_shards_fold", _lib.ptr(self._shard_record), 0, 0, 1, _lib.stream())
                self._shards.zero_()                                    # stream-ordered, behind whatever the aborted pass enqueued
        except Exception:
            pass

    def _pass_fold(self, *groups):
        """Add the shards of the named arena groups into the gradients and switch the shards off (every later kernel adds directly)."""
        if self._shards is None:
            return
        a, b = self.model.arena.range_of(*groups)
        _lib.call("clift_grad_shards_fold", _lib.ptr(self._shard_record), a - self._shard_range[0], b - a, 1, _lib.stream())

    def _jitter(self, n):
        """One U[0, 1) per ray (R:808-810), or None when perturb is 0.  ``host_rng``: from torch's CPU generator, exactly the reference's draw."""
        c = self.config
        if c.perturb == 0:
            return None
        if bool(getattr(c, "host_rng", False)):
            return (c.perturb * torch.rand(n, 1)).reshape(-1).to(self.device, non_blocking=True)
        j = torch.rand(n, device=self.device)
        return j if c.perturb == 1 else c.perturb * j

    def on_train_epoch_start(self, maintenance=True):
        """T:446-457, in the reference's order: the dist-reg weight ramps as lambda * (1 - exp(-0.25 epoch)); at ``bbox_aabb_reset_epochs`` the
        box shrinks to the alpha mask and the tables are cropped; at ``grid_upscale_epochs`` the tables are resampled to the next entry of the
        log-spaced voxel schedule, ``weight_decay`` drops to 0 for the rest of the run and optimizers + schedulers are rebuilt.  Pinned by golden
        G21 (the reference hook itself, five epochs).  A shrink in an epoch WITHOUT an upsample leaves the reference's optimizer holding the
        replaced table Parameters: the cropped tables then receive no updates until the next rebuild (G21 scenario B) -- reproduced, with a
        warning; it happens in no shipped config (every bbox_aabb_reset epoch is a grid_upscale epoch).
        ``maintenance=False``: only the ramp (the CLI resuming a checkpoint written in the middle of an epoch, whose tables already went
        through this epoch's hook)."""
        c = self.config
        e = self.current_epoch
        self.current_lambda_dist_reg = c.lambda_dist_reg * (1 - math.exp(-0.25 * e))
        if not maintenance:
            return
        shrink_at = [int(x) for x in (getattr(c, "bbox_aabb_reset_epochs", None) or ())]
        upscale_at = [int(x) for x in (getattr(c, "grid_upscale_epochs", None) or ())]
        if e in shrink_at:
            if self.renderer.update_bbox_aabb_and_shrink(self.model):
                if e not in upscale_at:
                    print(f"clift: epoch {e} shrinks the tables without a grid upsample: like the reference, the optimizers are not rebuilt and the "
                          "cropped tables stay fixed until the next grid_upscale epoch", flush=True)
                    self.setup_optimizers(carry=True)
        if e in upscale_at:
            import numpy as np
            voxels = torch.round(torch.exp(torch.linspace(np.log(c.min_grid_dim ** 3), np.log(c.max_grid_dim ** 3), len(upscale_at) + 1))).long().tolist()[1:]   # T:451
            target = self.renderer.get_target_resolution(voxels[upscale_at.index(e)])
            c.weight_decay = 0                                               # T:454
            self.model.upsample_volume_grid(target)
            self.renderer.update_step_size(target)
            self.setup_optimizers()                                          # T:457
            self.last_setup_epoch = e

    # ------------------------------------------------------------------ MultiStepLR (trainer/__init__.py:134-139, T:226-228)
    def _apply_lr_schedule(self):
        c = self.config
        if int(getattr(c, "warmup_epochs", 0) or 0) > 0:
            raise NotImplementedError("HotPathTrainer: warmup_epochs > 0 (GradualWarmupScheduler) is not built (0 in every shipped config)")
        steps = int(getattr(self, "sched_steps", 0))
        scale = float(getattr(c, "decay_gamma", 0.5)) ** sum(1 for m in (getattr(c, "decay_step", None) or ()) if steps >= int(m))
        self.opt_main.lr_scale = self.opt_inst.lr_scale = scale

    def scheduler_step(self):
        """T:226-228: both schedulers step once, at the last batch of every epoch.  The schedulers are re-created with the optimizers
        (``setup_optimizers``), so the milestones of ``decay_step`` count epochs since the last grid upsample (golden G21 scenario C,
        unpinned: it rests on Lightning's ``strategy.setup_optimizers`` replacing the scheduler objects)."""
        self.sched_steps = int(getattr(self, "sched_steps", 0)) + 1
        self._apply_lr_schedule()

    # ------------------------------------------------------------------ parameter groups in the reference's order (tensoRF.py:199-246)
    def torch_param_groups(self):
        m, c = self.model, self.config
        pl = lambda pre, kind: [f"{pre}_{kind}.{i}" for i in range(3)]
        seq = lambda prefix: [s.name for s in m.arena.slots if s.name.startswith(prefix + ".")]
        lg, ln = c.lr * 20, c.lr
        main = [(lg, pl("density", "line")), (lg, pl("appearance", "line")), (lg, pl("density", "plane")), (lg, pl("appearance", "plane")),
                (ln, ["appearance_basis_mat.weight"]), (ln, seq("render_appearance_mlp.mlp"))]
        if m.semantic_plane is not None:
            main += [(lg, pl("semantic", "plane")), (lg, pl("semantic", "line")), (ln, ["semantic_basis_mat.weight"])]
        main.append((ln, seq("render_semantic_mlp.mlp")))
        inst = []
        if m.instance_plane is not None:
            inst += [(lg, pl("instance", "plane")), (lg, pl("instance", "line")), (ln, ["instance_basis_mat.weight"])]
        inst.append((ln, seq("render_instance_mlp.mlp")))
        if m.slow_fast_mode and not bool(getattr(c, "use_DINO_style", True)):
            inst.append((ln, seq("render_instance_mlp.slow_mlp")))
        return main, inst

    def _allreduce(self, rng):
        if self.world > 1 or self.force_collectives:
            g = self.model.grad_flat[rng[0]:rng[1]]
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            g.mul_(1.0 / self.world)

    def _allreduce_start(self, rng):
        """Asynchronous form: the collective is queued behind what the current stream holds now and runs beside what is launched next."""
        g = self.model.grad_flat[rng[0]:rng[1]]
        work = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
        if self.allreduce_cu_reserve > 0 and self.device.type == "cuda":
            _lib.call("clift_set_cu_reserve", self.allreduce_cu_reserve)     # the launches that follow leave CUs to the collective
        return g, work

    def _allreduce_finish(self, started):
        g, work = started
        work.wait()
        if self.allreduce_cu_reserve > 0 and self.device.type == "cuda":
            _lib.call("clift_set_cu_reserve", 0)
        g.mul_(1.0 / self.world)

    # ------------------------------------------------------------------ overlap on/off, decided by measurement
    CAL_WARMUP, CAL_STEPS = 2, 4

    def _overlap_now(self):
        """Whether THIS main pass overlaps its exchange.  In "auto" mode the first CAL_WARMUP passes are not measured, the next
        CAL_STEPS run overlapped, the CAL_STEPS after that synchronously (each bracketed by a device synchronisation and timed on the
        host); the slower rank decides (MAX all-reduce of the medians), every rank takes the same branch afterwards."""
        if self.overlap_allreduce != "auto":
            return bool(self.overlap_allreduce)
        c = self._cal
        if c["decided"] is not None:
            return c["decided"]
        n = len(c["times"][True]) + len(c["times"][False]) + c.get("skipped", 0)
        return (n - self.CAL_WARMUP) < self.CAL_STEPS

    def _calibration_record(self, overlapped, seconds):
        c = self._cal
        if c["decided"] is not None or self.overlap_allreduce != "auto":
            return
        if c.get("skipped", 0) < self.CAL_WARMUP:
            c["skipped"] = c.get("skipped", 0) + 1
            return
        c["times"][bool(overlapped)].append(seconds)
        if len(c["times"][True]) >= self.CAL_STEPS and len(c["times"][False]) >= self.CAL_STEPS:
            med = lambda v: sorted(v)[len(v) // 2]
            t = torch.tensor([med(c["times"][True]), med(c["times"][False])], dtype=torch.float64,
                             device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c["overlap_ms"], c["sync_ms"] = float(t[0]) * 1e3, float(t[1]) * 1e3
            c["decided"] = bool(t[0] <= t[1])

    @property
    def overlap_decision(self):
        """None while undecided / not in auto mode; else dict(overlap=bool, overlap_ms=..., sync_ms=...)."""
        c = self._cal
        return None if c["decided"] is None else dict(overlap=c["decided"], overlap_ms=c["overlap_ms"], sync_ms=c["sync_ms"])

    # ------------------------------------------------------------------ sync-free capacity bookkeeping
    NOSYNC_WARMUP, NOSYNC_HEADROOM = 2, 1.3

    def _capacity(self, key, n_rays):
        """Capacity (rows of the compacted buffers) of a chunk of ``n_rays`` rays in pass ``key``, or None while that pass is still being
        learnt (the first steps synchronise like the default mode).  Before answering, absorb the asynchronous report of the previous
        step: true row count (the capacity follows its maximum with 30 % headroom) and overflow (count > capacity: that step dropped
        samples -- counted in ``overflow_steps`` and the capacity is raised at once)."""
        if not self.nosync:
            return None
        st = self._caps.setdefault((key, n_rays), {"seen": 0, "max": 0, "cap": None, "probe": None})
        pr = st["probe"]
        if pr is not None and pr[1].query():
            m, over = int(pr[0][0]), int(pr[0][1])
            st["probe"] = None
            st["max"] = max(st["max"], m, over)
            if over > 0:
                self.overflow_steps += 1
                engine.rows_limit(self.device)[1:2].zero_()
        if st["seen"] < self.NOSYNC_WARMUP:
            return None
        limit_all = n_rays * int(self.renderer.n_samples)
        st["cap"] = min(limit_all, max(4096, -(-int(self.NOSYNC_HEADROOM * st["max"]) // 4096) * 4096))
        return st["cap"]

    def _follow(self, key, n_rays, ctx):
        """After a chunk's forward: remember its row count -- read directly in the learning steps, through a pinned 8-byte copy + event
        (no wait) in the sync-free ones."""
        if not self.nosync:
            return
        st = self._caps[(key, n_rays)]
        st["seen"] += 1
        if not ctx.capped:
            st["max"] = max(st["max"], int(ctx.M))
        elif st["probe"] is None:
            host = torch.empty(2, dtype=torch.int32, pin_memory=True)
            host.copy_(engine.rows_limit(self.device), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st["probe"] = (host, ev)

    # ------------------------------------------------------------------ main pass (T:151-208)
    @_shard_guard
    def main_pass(self, batch, jitter=None, white_bg=None, lean=False, segments=None, segment_jitter=None):
        """batch: dict with rays (B,8), rgbs (B,3), probabilities (B,C), confidences (B,), mask (B,) bool/float.
        ``lean`` skips the instance heads, whose output the reference's main pass computes and discards (T:155).
        ``segments``: dict with rays (Bs,8), group (Bs,) int segment index, confidences (Bs,), n_groups -- the segment-consistency
        term of T:185-197 (batch[2]); it is part of the main loss, i.e. its gradient joins this pass's backward and Adam step."""
        c, m, r = self.config, self.model, self.renderer
        rays = batch["rays"]
        B = rays.shape[0]
        self._pass_begin(self.main_range)
        if jitter is None:
            jitter = self._jitter(B)
        chunk = c.chunk if c.chunk and c.chunk > 0 else B
        ctxs, outs = [], []
        for i in range(0, B, chunk):
            if white_bg is None:
                wb = self.white_bg or bool(torch.rand((1,)) < 0.5)       # renderer.py:164: one coin per chunk
            elif isinstance(white_bg, (list, tuple)):                    # (tests replaying the reference's recorded coins: one per chunk)
                wb = bool(white_bg[i // chunk])
            else:
                wb = bool(white_bg)
            n_c = min(chunk, B - i)
            o, ctx = engine.render_forward(m, r, rays[i:i + chunk], None if jitter is None else jitter[i:i + chunk], wb,
                                           want_inst=not lean, grad_heads=("app", "sem"), want_dist=False,        # T:155: the instance output is discarded
                                           cap=self._capacity("main", n_c))
            self._follow("main", n_c, ctx)
            ctxs.append(ctx)
            outs.append(o)
        rgb = outs[0]["rgb"] if len(outs) == 1 else torch.cat([o["rgb"] for o in outs], 0)
        sem = outs[0]["semantics"] if len(outs) == 1 else torch.cat([o["semantics"] for o in outs], 0)
        sem_on = self.current_epoch >= c.late_semantic_optimization
        w_rgb = float(c.lambda_rgb)
        w_sem = float(c.lambda_semantics) if sem_on else 0.0
        g_rgb = torch.empty_like(rgb)
        g_sem = torch.empty_like(sem)
        mask = batch.get("mask")
        maskf = mask.to(torch.float32) if mask is not None else None        # T:156-158 (masked pixels contribute nothing)
        self.losses.zero_()
        # T:177-182: "TTAConf" = soft targets x confidences; "NoTTAConf" = the label map as the target (class indices = one-hot rows for the
        # same kernel) x confidences; any other string = the label map as the target, NO confidences -- and since the mask reaches the
        # semantic term only through the zeroed confidences (T:158), masked pixels count in that form
        ce_mode = getattr(c, "probabilistic_ce_mode", "TTAConf")
        rgb_k, gt_k, conf_k, mask_k = rgb, batch["rgbs"], batch["confidences"], maskf
        if ce_mode != "TTAConf":
            batch = dict(batch)
            batch["probabilities"] = torch.nn.functional.one_hot(batch["semantics"].long(), sem.shape[1]).to(torch.float32)
            if ce_mode != "NoTTAConf":
                conf_k, mask_k = None, None
                if maskf is not None:       # T:156-157 by hand (both sides zeroed: a masked pixel's colour gradient is 0 either way)
                    rgb_k, gt_k = rgb * maskf[:, None], batch["rgbs"] * maskf[:, None]
        if getattr(c, "use_symmetric_ce", False):       # T:74-77: SCELoss(ce_alpha, ce_beta, weights) replaces the cross entropy
            _lib.call("clift_pixel_losses_sce", _lib.ptr(rgb_k), _lib.ptr(gt_k), _lib.ptr(sem), _lib.ptr(batch["probabilities"]),
                      _lib.ptr(conf_k), _lib.ptr(self.class_weights), _lib.ptr(mask_k), B, sem.shape[1], w_rgb, w_sem,
                      float(c.ce_alpha), float(c.ce_beta), _lib.ptr(self.losses), _lib.ptr(g_rgb), _lib.ptr(g_sem), _lib.stream())
        else:
            _lib.call("clift_pixel_losses", _lib.ptr(rgb_k), _lib.ptr(gt_k), _lib.ptr(sem), _lib.ptr(batch["probabilities"]),
                      _lib.ptr(conf_k), _lib.ptr(self.class_weights), _lib.ptr(mask_k), B, sem.shape[1], w_rgb, w_sem,
                      _lib.ptr(self.losses), _lib.ptr(g_rgb), _lib.ptr(g_sem), _lib.stream())
        gd = w_rgb * self.current_lambda_dist_reg / len(ctxs)
        if getattr(self, "_g_dist", (None, None))[0] != gd:          # (a device scalar that changes once per epoch: kept instead of refilled every step)
            self._g_dist = (gd, torch.full((1,), gd, dtype=torch.float32, device=self.device))
        g_dist = self._g_dist[1]
        gv = m.named_grad_views()
        seg_term = segments is not None and sem_on and float(getattr(c, "lambda_segment", 0.0)) != 0.0
        # Data-parallel runs: the TV term depends on the parameters only, so it goes FIRST (the scatter kernels accumulate on top of it),
        # and once the last chunk's head chains are issued everything but the density tables is final: that range (appearance tables +
        # both MLPs, ~3/4 of the bytes) is all-reduced under the density backward, the density tables after it.
        dp = self.world > 1 or self.force_collectives
        calibrating = dp and self.overlap_allreduce == "auto" and self._cal["decided"] is None and self.world > 1 and not seg_term
        early = dp and self._overlap_now() and not seg_term
        if calibrating:
            torch.cuda.synchronize(self.device)
            t_cal = time.perf_counter()
        started = []
        if early:
            self.losses[2] = m.total_tv_loss(None, c, self.current_epoch, accumulate_grad=True, scale=w_rgb)
        for k, ctx in enumerate(ctxs):
            s = slice(k * chunk, k * chunk + ctx.N)
            # (the head chains of the last chunk are issued: the MLP gradients are complete -- fold their shards before they travel)
            hook = (lambda: (self._pass_fold("net_app", "net_sem"), started.append(self._allreduce_start(self.early_range)))) \
                if (early and k == len(ctxs) - 1) else None
            engine.render_backward(m, ctx, gv, g_rgb[s], g_sem[s] if sem_on else None, None, g_dist, density_grad=True, before_density=hook)
        if seg_term:
            if getattr(c, "use_symmetric_ce", False):
                # the reference calls loss_semantics(features, CLASS INDICES) here (T:194); SCELoss takes log(labels) of them
                # (loss.py:53) -- that combination does not run in the reference either
                raise NotImplementedError("segment-consistency term with use_symmetric_ce: SCELoss is undefined on class-index targets")
            self._segment_term(segments, segment_jitter, gv, w_sem * float(c.lambda_segment))
        if not early:
            self._pass_fold("net_app", "net_sem")
            self.losses[2] = m.total_tv_loss(None, c, self.current_epoch, accumulate_grad=True, scale=w_rgb)
        if self.nosync:
            engine.reset_rows_limit(self.device)
        if early:
            self._allreduce(self.late_range)
            self._allreduce_finish(started[0])
        else:
            self._allreduce(self.main_range)
        if calibrating:
            torch.cuda.synchronize(self.device)
            self._calibration_record(early, time.perf_counter() - t_cal)
        self.opt_main.step(skip=() if sem_on else ("net_sem", "grid_sem"))      # no semantic term yet: the head's grad is None in the reference
        self.last_outputs = (rgb, sem)
        return ctxs

    def _segment_term(self, seg, jitter, gv, weight):
        """T:185-197: render the semantic features of the segment rays (renderer.forward_segment_feature), take the class of
        the per-segment mean feature, cross entropy of every ray against it (class weights, confidences); ``weight`` =
        lambda_semantics * lambda_segment scales the gradient that is accumulated into the semantic head."""
        c, m, r = self.config, self.model, self.renderer
        rays = seg["rays"]
        n = rays.shape[0]
        if n == 0:
            return
        if jitter is None:
            jitter = self._jitter(n)
        feats, ctx = engine.feature_forward(m, r, rays, jitter, "semantic", grad_heads=("sem",), cap=self._capacity("seg", n))
        self._follow("seg", n, ctx)
        C = feats.shape[1]
        G = int(seg["n_groups"])
        group = seg["group"].to(device=self.device, dtype=torch.int32).contiguous()
        conf = seg["confidences"].to(self.device, torch.float32).contiguous()
        work = torch.empty((G * C + G,), dtype=torch.float32, device=self.device)
        grad = torch.empty_like(feats)
        self.loss_segment = torch.zeros(1, dtype=torch.float32, device=self.device)
        _lib.call("clift_segment_loss", _lib.ptr(feats), feats.stride(0), _lib.ptr(group), _lib.ptr(conf), _lib.ptr(self.class_weights), n, C, G,
                  float(weight), _lib.ptr(work), _lib.ptr(self.loss_segment), _lib.ptr(grad), grad.stride(0), _lib.stream())
        engine.feature_backward(m, ctx, gv, grad)

    # ------------------------------------------------------------------ instance pass (T:210-222, 256-310)
    @_shard_guard
    def instance_pass(self, inst_batch, jitter=None):
        """inst_batch: list of dicts (one per image) with rays (n,8), instances (n,) int, confidences (n,)."""
        c, m, r = self.config, self.model, self.renderer
        if c.instance_loss_mode not in ("slow_fast", "contrastive", "linear_assignment"):
            raise NotImplementedError(f"HotPathTrainer: instance_loss_mode={c.instance_loss_mode!r} is not wired (slow_fast, contrastive and "
                                      "linear_assignment are; ae_loss is an unused experiment of the reference)")
        self._pass_begin(self.inst_range)
        gv = m.named_grad_views()
        contributed = False
        for img in inst_batch:
            rays = img["rays"]
            n = rays.shape[0]
            jit = jitter
            if jit is None:
                jit = self._jitter(n)
            (inst, xyz), ctx = engine.feature_forward(m, r, rays, jit, "instance", grad_heads=("fast",),     # slow half: detached (T:268)
                                                      cap=self._capacity("inst", n),
                                                      want_xyz=c.instance_loss_mode == "contrastive" and bool(getattr(c, "use_delta", False)))
            self._follow("inst", n, ctx)
            if c.instance_loss_mode == "slow_fast":
                # reference order: the features (fast and slow halves) are rendered first (T:214), THEN the EMA step of the slow
                # net at the top of the loss (T:258-259) -- the slow features of this step come from the pre-update weights
                # (golden G12).  One fused axpy over the contiguous fast/slow arena ranges; stream order keeps it behind the
                # forward's reads.
                f0, f1 = m.arena.range_of("inst_fast")
                s0, s1 = m.arena.range_of("inst_slow")
                assert f1 - f0 == s1 - s0, "arena groups inst_fast / inst_slow must have the same layout"
                _lib.call("clift_ema", _lib.ptr(m.param_flat[s0:s1]), _lib.ptr(m.param_flat[f0:f1]), f1 - f0, 0.9, _lib.stream())
                loss, g_inst = slow_fast_loss(inst, img["instances"], img["confidences"], return_grad=True)
            elif c.instance_loss_mode == "linear_assignment":
                # T:237-241: Hungarian-matched slots (host, like the reference), confidence-weighted cross entropy on the device.  An image
                # whose rays all sit on their slot already contributes the constant 0: no backward for it
                loss, g_inst = linear_assignment_loss(inst, img["instances"], img["confidences"], return_grad=True)
                if g_inst is None:
                    continue
            else:                                   # T:243-250: plain contrastive loss, optionally on points + features ("delta")
                use_delta = bool(getattr(c, "use_delta", False))
                if use_delta:
                    assert inst.shape[-1] == 3, "delta mode only works with 3D features"
                feats = xyz + inst if use_delta else inst
                loss, g_inst = contrastive_loss(feats, img["instances"], c.temperature, return_grad=True)
                if use_delta:                       # + 0.1 * mean ||delta||
                    nrm = torch.linalg.norm(inst, dim=-1, keepdim=True)
                    loss = loss + 0.1 * nrm.mean()
                    g_inst = g_inst + 0.1 * inst / (nrm.clamp_min(1e-30) * inst.shape[0])
            self.losses[3:4].add_(loss.reshape(1))
            engine.feature_backward(m, ctx, gv, g_inst, slow_grad=False)
            contributed = True
        if self.nosync:
            engine.reset_rows_limit(self.device)
        self._pass_fold("inst_fast")
        if c.instance_loss_mode == "linear_assignment":
            # reference: when the loss is the constant 0 every .grad stays None and torch's Adam skips every parameter, step counts and weight
            # decay included.  In a data-parallel run that holds only if NO rank contributed: one flag is summed over the ranks (this mode
            # synchronises with the host anyway, loss.create_virtual_gt_with_linear_assignment)
            if self.world > 1:
                flag = torch.tensor([1.0 if contributed else 0.0], device=self.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.SUM)
                contributed = bool(flag.item() > 0)
            if not contributed:
                return
        self._allreduce(self.inst_range)
        self.opt_inst.step()

    def training_step(self, batch, lean=None):
        """batch[0] = pixel batch dict, batch[1] = list of instance-image dicts (reference CombinedLoader layout)."""
        seg = batch.get(2) if (getattr(self.config, "segment_grouping_mode", "none") != "none" and
                               self.current_epoch >= self.config.segment_optimization_epoch) else None
        if lean is None:
            lean = bool(getattr(self.config, "skip_discarded_instance_heads", False))
        if not getattr(self.config, "optimize_instance_only", False):          # T:151: only the instance branch is optimised when set
            self.main_pass(batch[0], lean=lean, segments=seg)
        if self.current_epoch >= self.config.instance_optimization_epoch and batch.get(1):
            self.instance_pass(batch[1])

    # ------------------------------------------------------------------ checkpoint (Lightning layout)
    def state_dict_lightning(self):
        sd = {f"model.{k}": v for k, v in self.model.export_state_dict().items()}
        for k, v in self.renderer.state_dict().items():
            sd[f"renderer.{k}"] = v.detach().clone()
        sd["loss_semantics.weight"] = self.class_weights.detach().clone()
        return sd

    def checkpoint_dict(self, global_step=0, epoch_complete=True):
        """Lightning-layout checkpoint: ``state_dict`` (model.* / renderer.* / loss_semantics.weight), ``epoch``, ``global_step``,
        ``optimizer_states`` = the two Adam states in torch.optim.Adam's OWN state_dict layout with the reference's parameter-group order
        (tensoRF.py:199-246) and ``lr_schedulers`` = the two MultiStepLR states -- what ``trainer.fit(ckpt_path=...)`` of the reference
        restores -- plus a ``clift`` record for an exact continuation here (was the epoch finished; generators; frozen ranges)."""
        main, inst = self.torch_param_groups()
        c = self.config
        sched = lambda base: {"milestones": {int(m): 1 for m in (getattr(c, "decay_step", None) or ())}, "gamma": float(getattr(c, "decay_gamma", 0.5)),
                              "base_lrs": [float(lr) for lr, _ in base], "last_epoch": int(self.sched_steps), "verbose": False,
                              "_step_count": int(self.sched_steps) + 1, "_get_lr_called_within_step": False,
                              "_last_lr": [float(lr * self.opt_main.lr_scale) for lr, _ in base]}
        return {"state_dict": self.state_dict_lightning(), "epoch": self.current_epoch, "global_step": global_step,
                "pytorch-lightning_version": "2.0.4",
                "optimizer_states": [self.opt_main.torch_state_dict(main), self.opt_inst.torch_state_dict(inst)],
                "lr_schedulers": [sched(main), sched(inst)],
                "clift": {"epoch_complete": bool(epoch_complete), "last_setup_epoch": int(getattr(self, "last_setup_epoch", 0)),
                          "frozen": [sorted(self.opt_main.frozen), sorted(self.opt_inst.frozen)], "rng": self._rng_state()}}

    def save_checkpoint(self, path, global_step=0, epoch_complete=True):
        torch.save(self.checkpoint_dict(global_step, epoch_complete), path)

    def on_load_checkpoint(self, checkpoint):
        """T:461-470 followed by what Lightning does with the rest of the checkpoint (``load_state_dict`` of the module, then the optimizer
        and scheduler states): if the checkpoint's epoch is past a grid upsample the tables are brought to the checkpoint's resolution, the
        box and step size follow, ``weight_decay`` becomes 0 and the optimizers are rebuilt; then every weight, the renderer buffers, both
        Adam states and the scheduler position are restored.  Accepts checkpoints of the reference (torch-layout ``optimizer_states``) and of
        earlier builds of this repo (flat ``m`` / ``v`` records).  Golden G21 pins the hook and the continuation."""
        c, m, r = self.config, self.model, self.renderer
        dev = self.device
        sd = checkpoint["state_dict"]
        for epoch in [int(x) for x in (getattr(c, "grid_upscale_epochs", None) or ())][::-1]:
            if checkpoint["epoch"] >= epoch:
                c.weight_decay = 0                                                 # T:468
                break
        grid = [int(x) for x in sd["renderer.grid_dim"].tolist()]
        have = [int(x) for x in r.grid_dim.tolist()]
        if grid != have:            # (T:463-466 upsamples only past an upscale epoch; in every shipped config a changed grid implies one.  The shapes must
            m.upsample_volume_grid(grid)                                           #  match for the strict load below either way; the values are overwritten)
        missing, unexpected = m.load_state_dict({k[len("model."):]: v.to(dev) for k, v in sd.items() if k.startswith("model.")}, strict=True)
        r.bbox_aabb.data = sd["renderer.bbox_aabb"].to(dev)
        r.update_step_size(grid)
        if "loss_semantics.weight" in sd:
            self.class_weights = sd["loss_semantics.weight"].to(dev, torch.float32).clone()
        self.current_epoch = int(checkpoint["epoch"])
        self.setup_optimizers()                                                    # T:469, in the (possibly resized) arena layout
        self.on_train_epoch_start(maintenance=False)
        ost = checkpoint.get("optimizer_states")
        if ost and len(ost) == 2:
            if "param_groups" in ost[0]:
                main, inst = self.torch_param_groups()
                self.opt_main.load_torch_state_dict(ost[0], main)
                self.opt_inst.load_torch_state_dict(ost[1], inst)
            elif "m" in ost[0] and ost[0]["m"].numel() == self.opt_main.m.numel():
                self.opt_main.load_state_dict(ost[0])
                self.opt_inst.load_state_dict(ost[1])
        extra = checkpoint.get("clift", {})
        if checkpoint.get("lr_schedulers"):
            self.sched_steps = int(checkpoint["lr_schedulers"][0]["last_epoch"])
        elif "last_setup_epoch" in extra:                                          # earlier builds: milestones counted from the last rebuild
            done = self.current_epoch + (1 if extra.get("epoch_complete", True) else 0)
            self.sched_steps = max(0, done - int(extra["last_setup_epoch"]))
        self._apply_lr_schedule()
        for o, fz in zip((self.opt_main, self.opt_inst), extra.get("frozen", ((), ()))):
            o.frozen = set(fz)
        self.last_setup_epoch = int(extra.get("last_setup_epoch", 0))
        return extra

    # ------------------------------------------------------------------ validation (T:356-400)
    @torch.no_grad()
    def validation_step(self, batch, things, stuff, faulty_classes=(0,)):
        """One validation view through the reference's ``validation_step``: masked MSE + PSNR, the semantic loss of the configured mode, mIoU
        against the (machine-generated) training labels with class 0 ignored, PQ / SQ / RQ of (semantic argmax, instance argmax), and the same
        against the ground-truth ``rs_*`` labels.  ``batch``: rays (P, 8), rgbs, semantics, instances, mask, rs_semantics, rs_instances,
        probabilities, confidences.  Returns the reference's ``metrics_data`` dict (11 floats).  Host-side evaluation: the render is the HIP
        path, the metrics are numpy / torch on its outputs (golden G21: ``A.val.*``)."""
        from .inference import ConfusionMatrix, psnr, render_rays
        from .metrics import panoptic_quality
        from . import loss as L
        c, dev = self.config, self.device
        d = lambda k: batch[k].to(dev)
        mask = d("mask").bool()
        rgb, sem, inst, _depth = render_rays(self.model, self.renderer, d("rays"), c.chunk, self.white_bg)
        rgb, sem = rgb.clone(), sem.clone()
        rgbs = d("rgbs").clone()
        rgb[~mask] = 0
        rgbs[~mask] = 0
        loss_rgb = torch.mean((rgb - rgbs) ** 2)
        metric_psnr = psnr(rgb, rgbs)
        sem[~mask] = 0
        semantics = d("semantics").long()
        if getattr(c, "use_symmetric_ce", False):
            lossf = L.SCELoss(float(c.ce_alpha), float(c.ce_beta), self.class_weights)
        else:
            lossf = torch.nn.CrossEntropyLoss(reduction="none", weight=self.class_weights)
        mode = getattr(c, "probabilistic_ce_mode", "TTAConf")
        if mode == "TTAConf":
            loss_sem = (lossf(sem, d("probabilities")) * d("confidences")).mean()
        elif mode == "NoTTAConf":
            loss_sem = (lossf(sem, semantics) * d("confidences")).mean()
        else:
            loss_sem = lossf(sem, semantics).mean()
        C = self.model.num_semantic_classes
        pred_valid = sem.argmax(1)
        pred_all = pred_valid.clone()
        pred_valid[semantics == 0] = 0
        inst_arg = inst.argmax(1)
        iou = ConfusionMatrix(num_classes=C, ignore_class=[0]).add_batch(pred_valid.cpu().numpy(), semantics.cpu().numpy(), return_miou=True)
        pq, sq, rq = panoptic_quality(torch.stack([pred_valid, inst_arg], 1), torch.stack([semantics, d("instances").long()], 1), things, stuff,
                                      allow_unknown_preds_category=True)
        rs_sem, rs_inst = d("rs_semantics").long(), d("rs_instances").long()
        rs_iou = ConfusionMatrix(num_classes=C, ignore_class=list(faulty_classes)).add_batch(pred_all.cpu().numpy(), rs_sem.cpu().numpy(), return_miou=True)
        rs_pq, rs_sq, rs_rq = panoptic_quality(torch.stack([pred_all, inst_arg], 1), torch.stack([rs_sem, rs_inst], 1), things, stuff,
                                               allow_unknown_preds_category=True)
        return {"loss_rgb": float(loss_rgb), "loss_sem": float(loss_sem), "psnr": float(metric_psnr), "iou": float(iou), "pq": float(pq), "sq": float(sq),
                "rq": float(rq), "rs_iou": float(rs_iou), "rs_pq": float(rs_pq), "rs_sq": float(rs_sq), "rs_rq": float(rs_rq)}

    def _rng_state(self):
        """Generators the step draws from: torch's CPU generator (white-background coin, R:164), the device's default generator
        (jitter, ray subsampling) and the CLI's pixel-batch generator when it registered one (``self.pixel_generator``)."""
        st = {"cpu": torch.get_rng_state()}
        if self.device.type == "cuda":
            st["cuda"] = torch.cuda.get_rng_state(self.device)
        g = getattr(self, "pixel_generator", None)
        if g is not None:
            st["pixel"] = g.get_state()
        return st

    def load_rng_state(self, st):
        if not st:
            return
        torch.set_rng_state(st["cpu"])
        if "cuda" in st and self.device.type == "cuda":
            torch.cuda.set_rng_state(st["cuda"], self.device)
        g = getattr(self, "pixel_generator", None)
        if g is not None and "pixel" in st:
            g.set_state(st["pixel"])
