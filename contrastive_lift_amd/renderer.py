"""TensoRFRenderer -- drop-in mirror of reference model/renderer/panopli_tensoRF_renderer.py:37-300,626-634,756-761.

Same constructor, buffers (``bbox_aabb``, ``grid_dim``, ``inv_box_extent``, ``units``), python attributes
(``step_size``, ``n_samples``) and call signatures; the per-ray arithmetic runs in libclift.so via engine.py.
``forward`` is differentiable through a single ``torch.autograd.Function`` whose backward launches the HIP
backward kernels; the trainer (trainer.py) bypasses autograd and drives engine.py directly.
"""
import torch
from torch import nn

from . import engine


class _RenderFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, model, renderer, rays, jitter, white_bg, names, *params):
        out, rctx = engine.render_forward(model, renderer, rays, jitter, white_bg)
        ctx.model, ctx.rctx, ctx.names = model, rctx, names
        ctx.needs = [p.requires_grad for p in params]
        ctx.mark_non_differentiable(out["depth"])
        inst = out["instances"]
        if inst is None:
            inst = torch.zeros((rays.shape[0], 0), dtype=torch.float32, device=rays.device)
        return out["rgb"], out["semantics"], inst, out["depth"], out["dist_reg"]

    @staticmethod
    def backward(ctx, g_rgb, g_sem, g_inst, _g_depth, g_dist):
        model = ctx.model
        grads = _alloc_grads(model)
        slow_grad = any(n.startswith("render_instance_mlp.slow_mlp") and need for n, need in zip(ctx.names, ctx.needs))
        gd = g_dist.reshape(1).to(torch.float32).contiguous() if g_dist is not None else None
        engine.render_backward(model, ctx.rctx, grads, g_rgb, g_sem, g_inst if g_inst.shape[1] > 0 else None, gd,
                               density_grad=True, slow_grad=slow_grad)
        return (None,) * 6 + tuple(grads[n] if need else None for n, need in zip(ctx.names, ctx.needs))


class _FeatureFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, model, renderer, rays, jitter, head, names, *params):
        out, rctx = engine.feature_forward(model, renderer, rays, jitter, head)
        ctx.model, ctx.rctx, ctx.names, ctx.head = model, rctx, names, head
        ctx.needs = [p.requires_grad for p in params]
        if head == "instance":
            ctx.mark_non_differentiable(out[1])
            return out
        return out

    @staticmethod
    def backward(ctx, g_out, *_):
        model = ctx.model
        grads = _alloc_grads(model)
        slow_grad = any(n.startswith("render_instance_mlp.slow_mlp") and need for n, need in zip(ctx.names, ctx.needs))
        engine.feature_backward(model, ctx.rctx, grads, g_out, slow_grad=slow_grad)
        # what a feature pass reaches: the head's MLP and, when the head sits on its own VM grid (tensoRF.py:70-83), that grid's tables and basis
        pref = (("render_semantic_mlp", "semantic_plane", "semantic_line", "semantic_basis_mat") if ctx.head == "semantic" else
                ("render_instance_mlp", "instance_plane", "instance_line", "instance_basis_mat"))
        return (None,) * 6 + tuple(grads[n] if (need and n.startswith(pref)) else None for n, need in zip(ctx.names, ctx.needs))


def _alloc_grads(model):
    """Zeroed gradient tensors with exactly the parameters' memory layout (fresh arena buffer)."""
    return model.arena.views(model.arena.new_buffer())


class TensoRFRenderer(nn.Module):

    def __init__(self, bbox_aabb, grid_dim, stop_semantic_grad=True, semantic_weight_mode="none", step_ratio=0.5,
                 distance_scale=25, raymarch_weight_thres=0.0001, alpha_mask_threshold=0.0075, parent_renderer_ref=None,
                 instance_id=0, feature_stop_grad=False):
        super().__init__()
        self.register_buffer("bbox_aabb", torch.as_tensor(bbox_aabb, dtype=torch.float32).clone())
        self.register_buffer("grid_dim", torch.LongTensor(list(grid_dim)))
        self.register_buffer("inv_box_extent", torch.zeros([3]))
        self.register_buffer("units", torch.zeros([3]))
        # "softmax" (R:160) and "argmax" (R:142) are compared for; any other string renders as "none", as in the reference
        self.semantic_weight_mode = semantic_weight_mode
        self.parent_renderer_ref = parent_renderer_ref
        self.step_ratio = step_ratio
        self.distance_scale = distance_scale
        self.raymarch_weight_thres = raymarch_weight_thres
        self.alpha_mask_threshold = alpha_mask_threshold
        self.step_size = None
        self.n_samples = None
        self.stop_semantic_grad = stop_semantic_grad
        self.feature_stop_grad = feature_stop_grad
        self.instance_id = instance_id
        self.update_step_size(self.grid_dim)

    # ------------------------------------------------------------------ host scalars (renderer.py:59-78)
    def _refresh_host(self):
        aabb = self.bbox_aabb.detach().to("cpu", torch.float32)
        self.bbox_aabb_host = (aabb[0].tolist(), aabb[1].tolist())
        self.inv_box_extent_host = self.inv_box_extent.detach().to("cpu", torch.float32).tolist()
        self.step_size_host = float(self.step_size)

    def update_step_size(self, grid_dim):
        if isinstance(grid_dim, (tuple, list)):
            grid_dim = torch.tensor([int(g) for g in grid_dim], dtype=torch.int64)
        dev = self.bbox_aabb.device
        aabb = self.bbox_aabb.detach().to("cpu", torch.float32)
        g = grid_dim.detach().to("cpu", torch.int64)
        ext = aabb[1] - aabb[0]
        self.grid_dim.data = g.to(dev)
        self.inv_box_extent.data = (2.0 / ext).to(dev)
        units = ext / (g - 1 + 1e-3)
        self.units.data = units.to(dev)
        self.step_size = torch.mean(units) * self.step_ratio
        diag = torch.sqrt(torch.sum(torch.square(ext)))
        self.n_samples = int((diag / self.step_size).item()) + 1
        self._refresh_host()

    def update_step_ratio(self, step_ratio):
        self.step_ratio = step_ratio
        units = self.units.detach().to("cpu", torch.float32)
        aabb = self.bbox_aabb.detach().to("cpu", torch.float32)
        self.step_size = torch.mean(units) * self.step_ratio
        diag = torch.sqrt(torch.sum(torch.square(aabb[1] - aabb[0])))
        self.n_samples = int((diag / self.step_size).item()) + 1
        self._refresh_host()

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        if self.step_size is not None:
            self._refresh_host()
        return r

    def get_target_resolution(self, n_voxels):
        """renderer.py:756-761."""
        aabb = self.bbox_aabb.detach().to("cpu", torch.float32)
        ext = aabb[1] - aabb[0]
        voxel = (ext.prod() / n_voxels).pow(1 / 3)
        return tuple(max(int(x), 1) for x in (ext / voxel).long().tolist())

    def normalize_coordinates(self, xyz_sampled):
        return (xyz_sampled - self.bbox_aabb[0]) * self.inv_box_extent - 1

    # ------------------------------------------------------------------ randomness (same CPU draws as the reference)
    @staticmethod
    def _draw_jitter(n_rays, perturb, is_train, device):
        """renderer.py:808-810: one U[0,1) per ray from the CPU generator, scaled by perturb; None if not training."""
        if is_train and perturb != 0:
            return (perturb * torch.rand(n_rays, 1)).reshape(-1).to(device, non_blocking=True)
        return None

    # ------------------------------------------------------------------ forward passes
    def forward(self, tensorf, rays, perturb, white_bg, is_train, jitter=None, white_bg_resolved=None):
        """renderer.py:80-176.  Returns (rgb (N,3), semantics (N,C), instances (N,D), depth (N,), feats (1,1),
        dist_regularizer ()).  ``jitter`` / ``white_bg_resolved`` override the two random draws (tests, trainer)."""
        if jitter is None:
            jitter = self._draw_jitter(rays.shape[0], perturb, is_train, rays.device)
        if white_bg_resolved is None:
            white_bg_resolved = bool(white_bg) or bool(is_train and torch.rand((1,)) < 0.5)     # renderer.py:164
        names = [s.name for s in tensorf.arena.slots]
        params = [tensorf.get_parameter(n) for n in names]
        rgb, sem, inst, depth, dist_reg = _RenderFn.apply(tensorf, self, rays, jitter, white_bg_resolved, names, *params)
        feats = torch.zeros([1, 1], device=rays.device)
        return rgb, sem, inst, depth, feats, dist_reg

    def forward_instance_feature(self, tensorf, rays, perturb, is_train, jitter=None):
        """renderer.py:178-217 -> (instance_map (N,D), points_xyz (N,3))."""
        if jitter is None:
            jitter = self._draw_jitter(rays.shape[0], perturb, is_train, rays.device)
        names = [s.name for s in tensorf.arena.slots]
        params = [tensorf.get_parameter(n) for n in names]
        return _FeatureFn.apply(tensorf, self, rays, jitter, "instance", names, *params)

    def forward_segment_feature(self, tensorf, rays, perturb, is_train, jitter=None):
        """renderer.py:259-300 -> segment_map (N,C) (log-probabilities in softmax mode)."""
        if jitter is None:
            jitter = self._draw_jitter(rays.shape[0], perturb, is_train, rays.device)
        names = [s.name for s in tensorf.arena.slots]
        params = [tensorf.get_parameter(n) for n in names]
        return _FeatureFn.apply(tensorf, self, rays, jitter, "semantic", names, *params)

    # ------------------------------------------------------------------ alpha-mask shrink (renderer.py:669-754)
    @torch.no_grad()
    def compute_alpha(self, tensorf, xyz_locs, step_size):
        xyz_sampled = self.normalize_coordinates(xyz_locs)
        sigma = tensorf.compute_density(xyz_sampled.reshape(-1, 3)).reshape(xyz_locs.shape[:-1])
        return 1 - torch.exp(-sigma * step_size)

    @torch.no_grad()
    def get_dense_alpha(self, tensorf):
        """alpha on the (Rx,Ry,Rz) lattice of the current box (renderer.py:717-729): one clift_density_points launch
        over the whole lattice instead of a Python loop over x-slabs."""
        dev = self.bbox_aabb.device
        g = [int(x) for x in self.grid_dim.tolist()]
        samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, g[0]), torch.linspace(0, 1, g[1]), torch.linspace(0, 1, g[2]),
                                             indexing='ij'), -1).to(dev)
        dense_xyz = self.bbox_aabb[0] * (1 - samples) + self.bbox_aabb[1] * samples
        alpha = self.compute_alpha(tensorf, dense_xyz.reshape(-1, 3), self.step_size.to(dev)).view(g[0], g[1], g[2])
        return alpha, dense_xyz

    @torch.no_grad()
    def occupied_index_box(self, tensorf):
        """Index bounding box of the lattice voxels whose 3^3-max-pooled alpha reaches ``alpha_mask_threshold`` (the first half of
        renderer.py:669-680), computed by ONE library call (clift_alpha_bbox: lattice alpha, pool, threshold, min/max reduction
        on the device).  Returns (lo_idx, hi_idx, n_voxels) as Python ints per axis, or None when no voxel qualifies."""
        import ctypes as C
        from . import _lib
        dev = self.bbox_aabb.device
        g = [int(x) for x in self.grid_dim.tolist()]
        ticks = [torch.linspace(0, 1, n).to(dev) for n in g]                 # lattice parameters per axis, as R:718-722
        views = tensorf.named_views()
        vd = engine.vm_struct(views, "density", engine.grid_res(views))
        lo, hi = self.bbox_aabb_host
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        alpha = torch.empty(g[0] * g[1] * g[2], dtype=torch.float32, device=dev)
        box = torch.empty(7, dtype=torch.int32, device=dev)
        _lib.call("clift_alpha_bbox", C.byref(vd), f3(lo), f3(hi), f3(self.inv_box_extent_host), _lib.ptr(ticks[0]), _lib.ptr(ticks[1]),
                  _lib.ptr(ticks[2]), g[0], g[1], g[2], float(tensorf.splus_density_shift), float(self.step_size_host),
                  float(self.alpha_mask_threshold), _lib.ptr(alpha), _lib.ptr(box), _lib.stream())
        box = box.tolist()
        if box[6] == 0:
            return None
        return box[0:3], box[3:6], box[6], ticks

    @torch.no_grad()
    def update_bbox_aabb_and_shrink(self, tensorf, fractional_lenience=1.0):
        """renderer.py:668-715.  The device finds the occupied index box; what is left for the host is three-vector arithmetic:
        lattice coordinates of the two corners, the lenience scaling about the centre, clipping to the current (and the parent's)
        box, and rounding the new corners to voxel indices for the crop."""
        found = self.occupied_index_box(tensorf)
        if found is None:
            return False
        lo_idx, hi_idx, _, ticks = found
        box_lo, box_hi = self.bbox_aabb[0], self.bbox_aabb[1]

        def lattice(idx):                 # coordinates of a lattice node: the same two-product blend that built the lattice (R:723)
            t = torch.stack([ticks[a][idx[a]] for a in range(3)])
            return box_lo * (1 - t) + box_hi * t
        corner_lo, corner_hi = lattice(lo_idx), lattice(hi_idx)       # monotone per axis: min / max coordinate = node at min / max index
        centre, half = (corner_lo + corner_hi) / 2, ((corner_hi - corner_lo) * fractional_lenience) / 2
        new_lo, new_hi = torch.maximum(box_lo, centre - half), torch.minimum(box_hi, centre + half)
        if self.parent_renderer_ref is not None:
            new_lo = torch.maximum(self.parent_renderer_ref.bbox_aabb[0], new_lo)
            new_hi = torch.minimum(self.parent_renderer_ref.bbox_aabb[1], new_hi)
        # voxel range to keep: first = round(offset / unit), one past last = min(round(offset / unit) + 1, grid size)
        first = torch.round((new_lo - box_lo) / self.units).long()
        end = torch.minimum(torch.round((new_hi - box_lo) / self.units).long() + 1, self.grid_dim)
        kept = end - first
        if not bool((kept > 0).all()):
            return False
        tensorf.shrink(first.tolist(), end.tolist())
        self.bbox_aabb.data = torch.stack((new_lo, new_hi))
        self.update_step_size(tuple(int(x) for x in kept.tolist()))
        return True

    @staticmethod
    def raw_to_alpha(sigma, dist):
        """renderer.py:626-631 (plain torch; the fused path lives in clift_march_fwd)."""
        alpha = 1. - torch.exp(-sigma * dist)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1. - alpha + 1e-10], -1), -1)
        return alpha, alpha * T[..., :-1], T[..., -1:]

    @property
    def extent(self):
        return self.bbox_aabb[1] - self.bbox_aabb[0]

    @property
    def position(self):
        return (self.bbox_aabb[0] + self.bbox_aabb[1]) / 2
