"""TensorVMSplit -- drop-in mirror of the reference radiance field (model/radiance_field/tensoRF.py:32-315).

Same constructor signature, attribute names, ``state_dict`` keys/shapes and optimizer param groups as the
reference; the arithmetic runs in libclift.so (HIP, gfx950).  Storage differs from the reference on purpose:
every trainable tensor is a view into one flat arena (arena.py) -- VM tables channels-last, weight matrices
with a 16-byte row pitch -- while keeping the reference's logical shapes.

Both head arrangements of the reference are built: xyz MLPs (``use_semantic_mlp`` / ``use_instance_mlp`` True: the template default and the
contrastive-lift configs) and heads on their own VM grids (False: 3 x 32 components -> basis Linear -> 27 features -> 3-layer MLP,
tensoRF.py:70-83; the allgrid / instGRIDsemMLP / onlyRGBsegGRID overlays, round 5).  Distilled-feature grids (tensoRF.py:91-94) raise
NotImplementedError.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib
from .arena import Arena, Slot

MATRIX_MODE = [[0, 1], [0, 2], [1, 2]]   # tensoRF.py:61
VECTOR_MODE = [2, 1, 0]                  # tensoRF.py:62


def _seq_mlp(dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class MLPRenderFeature(nn.Module):
    """Appearance head container (tensoRF.py:383-418): [feat, dirs, PE(feat), PE(dirs)] -> 128 -> 128 -> 3 -> sigmoid."""

    def __init__(self, in_channels, out_channels=3, pe_view=2, pe_feat=2, dim_mlp_color=128):
        super().__init__()
        self.pe_view, self.pe_feat = pe_view, pe_feat
        self.in_channels = in_channels
        self.output_channels = out_channels
        self.in_feat_mlp = 2 * pe_view * 3 + 2 * pe_feat * in_channels + in_channels + 3
        self.mlp = _seq_mlp([self.in_feat_mlp, dim_mlp_color, dim_mlp_color, out_channels])
        nn.init.constant_(self.mlp[-1].bias, 0)

    def forward(self, viewdirs, features):
        from .engine import appearance_mlp_points
        return appearance_mlp_points(self, viewdirs, features)


class MLPRenderSemanticFeature(nn.Module):
    """Semantic head container (tensoRF.py:565-594): xyz -> 256 x (L-1) -> C, output activation softmax|identity."""

    def __init__(self, in_channels, out_channels, num_mlp_layers=5, dim_mlp=256, softmax=True):
        super().__init__()
        self.output_channels = out_channels
        self.softmax = softmax
        self.in_feat_mlp = in_channels
        self.mlp = _seq_mlp([in_channels] + [dim_mlp] * (num_mlp_layers - 1) + [out_channels])

    def forward(self, distilled_feats, feat_xyz):
        from .engine import feat_mlp_points, xyz_mlp_points
        out = (xyz_mlp_points if self.in_feat_mlp == 3 else feat_mlp_points)(self.mlp, feat_xyz)
        return torch.softmax(out, -1) if self.softmax else out


class MLPRenderInstanceFeature(nn.Module):
    """Instance head container (tensoRF.py:462-511): fast ``mlp`` and (slow_fast_mode) EMA ``slow_mlp``."""

    def __init__(self, in_channels, out_channels, num_mlp_layers=4, dim_mlp=256, slow_fast_mode=False):
        super().__init__()
        self.output_channels = out_channels
        self.slow_fast_mode = slow_fast_mode
        self.in_feat_mlp = in_channels
        dims = [in_channels] + [dim_mlp] * (num_mlp_layers - 1) + [out_channels]
        self.mlp = _seq_mlp(dims)
        if slow_fast_mode:
            self.slow_mlp = _seq_mlp(dims)     # same architecture, independently initialised (tensoRF.py:483-491)

    def forward(self, distilled_feats, feat_xyz):
        from .engine import feat_mlp_points, xyz_mlp_points
        fn = xyz_mlp_points if self.in_feat_mlp == 3 else feat_mlp_points
        out = fn(self.mlp, feat_xyz)
        if self.slow_fast_mode:
            out = torch.cat([out, fn(self.slow_mlp, feat_xyz)], -1)
        return out


class TensorVMSplit(nn.Module):

    def __init__(self, grid_dim, num_density_comps=(16, 16, 16), num_appearance_comps=(48, 48, 48), num_semantics_comps=None,
                 num_instance_comps=None, dim_appearance=27, dim_semantics=27, dim_instances=27, splus_density_shift=-10,
                 pe_view=2, pe_feat=2, dim_mlp_color=128, dim_mlp_semantics=128, dim_mlp_instance=256,
                 num_semantic_classes=0, dim_feature_instance=None, output_mlp_semantics=torch.nn.Softmax(dim=-1),
                 use_semantic_mlp=False, use_instance_mlp=False, use_feature_reg=False,
                 use_distilled_features_semantic=False, use_distilled_features_instance=False,
                 num_feature_comps=(48, 48, 48), pe_sem=0, pe_ins=0, slow_fast_mode=False, use_proj=False, device=None):
        super().__init__()
        for flag, comps, what in ((use_semantic_mlp, num_semantics_comps, "semantic"), (use_instance_mlp, num_instance_comps, "instance")):
            if not flag and comps is not None and (len(set(comps)) != 1 or comps[0] not in (16, 32, 48)):
                raise NotImplementedError(f"clift: the {what} grid takes 16, 32 or 48 components per plane, equal on the three planes (got {comps})")
        if use_distilled_features_semantic or use_distilled_features_instance or use_proj or use_feature_reg:
            raise NotImplementedError("clift: distilled-feature grids / projection head / feature regulariser are off "
                                      "in every shipped contrastive-lift config and are not built")
        for flag, width, what in ((use_semantic_mlp, dim_mlp_semantics, "dim_mlp_semantics"), (use_instance_mlp, dim_mlp_instance, "dim_mlp_instance")):
            if not flag and width % 4 != 0:      # the hidden activations of a grid head's MLP are the next layer's GEMM operand: rows must be 16-byte multiples
                raise NotImplementedError(f"clift: {what} of a head on its own VM grid must be a multiple of 4 (got {width})")
        if pe_sem != 0 or pe_ins != 0:
            raise NotImplementedError("clift: pe_sem / pe_ins > 0 not built (0 in every shipped config)")
        if len(set(num_density_comps)) != 1 or len(set(num_appearance_comps)) != 1:
            raise NotImplementedError("clift: per-plane component counts must be equal")
        self.num_density_comps, self.num_appearance_comps = tuple(num_density_comps), tuple(num_appearance_comps)
        self.num_semantics_comps, self.num_instance_comps = num_semantics_comps, num_instance_comps
        self.dim_appearance, self.dim_semantics, self.dim_instances = dim_appearance, dim_semantics, dim_instances
        self.dim_feature_instance = dim_feature_instance
        ins_out = dim_feature_instance // 2 if slow_fast_mode else dim_feature_instance
        self.num_semantic_classes = num_semantic_classes
        self.splus_density_shift = splus_density_shift
        self.use_semantic_mlp, self.use_instance_mlp = use_semantic_mlp, use_instance_mlp
        self.slow_fast_mode, self.use_proj, self.use_feature_reg = slow_fast_mode, use_proj, False
        self.use_distilled_features_semantic = self.use_distilled_features_instance = False
        self.pe_view, self.pe_feat, self.dim_mlp_color = pe_view, pe_feat, dim_mlp_color
        self.matrix_mode, self.vector_mode = MATRIX_MODE, VECTOR_MODE
        self.semantic_plane = self.semantic_line = self.semantic_basis_mat = None
        self.instance_plane = self.instance_line = self.instance_basis_mat = None
        self.feature_plane = self.feature_line = self.feature_basis_mat = self.render_feature_mlp = None
        softmax = isinstance(output_mlp_semantics, torch.nn.Softmax)
        if not softmax and not isinstance(output_mlp_semantics, torch.nn.Identity):
            raise NotImplementedError("clift: output_mlp_semantics must be Softmax(dim=-1) or Identity")
        # construction order follows the reference __init__ (tensoRF.py:63-85) so that a given torch seed
        # produces the same initial weights as the reference
        self.density_plane, self.density_line = self.init_one_svd(self.num_density_comps, grid_dim, 0.1)
        self.appearance_plane, self.appearance_line = self.init_one_svd(self.num_appearance_comps, grid_dim, 0.1)
        self.appearance_basis_mat = nn.Linear(sum(self.num_appearance_comps), dim_appearance, bias=False)
        self.render_appearance_mlp = MLPRenderFeature(dim_appearance, 3, pe_view, pe_feat, dim_mlp_color)
        self.render_instance_mlp = None
        if dim_feature_instance is not None:
            if num_instance_comps is not None and not use_instance_mlp:       # tensoRF.py:70-74: the head on its own VM grid
                self.instance_plane, self.instance_line = self.init_one_svd(tuple(num_instance_comps), grid_dim, 0.1)
                self.instance_basis_mat = nn.Linear(sum(num_instance_comps), dim_instances, bias=False)
                self.render_instance_mlp = MLPRenderInstanceFeature(dim_instances, ins_out, num_mlp_layers=3, dim_mlp=dim_mlp_instance,
                                                                    slow_fast_mode=slow_fast_mode)
            elif use_instance_mlp:
                self.render_instance_mlp = MLPRenderInstanceFeature(3, ins_out, num_mlp_layers=4, dim_mlp=dim_mlp_instance,
                                                                    slow_fast_mode=slow_fast_mode)
        self.render_semantic_mlp = None
        if num_semantics_comps is not None and not use_semantic_mlp:          # tensoRF.py:78-82
            self.semantic_plane, self.semantic_line = self.init_one_svd(tuple(num_semantics_comps), grid_dim, 0.1)
            self.semantic_basis_mat = nn.Linear(sum(num_semantics_comps), dim_semantics, bias=False)
            self.render_semantic_mlp = MLPRenderSemanticFeature(dim_semantics, num_semantic_classes, num_mlp_layers=3, dim_mlp=dim_mlp_semantics,
                                                                softmax=softmax)
        elif use_semantic_mlp:
            self.render_semantic_mlp = MLPRenderSemanticFeature(3, num_semantic_classes, softmax=softmax)
        if self.render_semantic_mlp is None:
            raise NotImplementedError("clift: a field without a semantic head (use_semantic_mlp False and no num_semantics_comps) is not built")
        self.arena = None
        self.param_flat = self.grad_flat = None
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.pack(dev)

    # ------------------------------------------------------------------ construction helpers
    def init_one_svd(self, n_components, grid_resolution, scale):
        plane, line = [], []
        for i in range(3):
            v = VECTOR_MODE[i]
            a, b = MATRIX_MODE[i]
            plane.append(nn.Parameter(scale * torch.randn((1, n_components[i], grid_resolution[b], grid_resolution[a]))))
            line.append(nn.Parameter(scale * torch.randn((1, n_components[i], grid_resolution[v], 1))))
        return nn.ParameterList(plane), nn.ParameterList(line)

    def _layout(self):
        """(name, owner module/list, attribute or index, kind, group) in arena order = optimizer-group order."""
        out = []
        for pre, grp in (("density", "grid_density"), ("appearance", "grid_app")):
            for i in range(3):
                out.append((f"{pre}_plane.{i}", getattr(self, f"{pre}_plane"), i, "grid", grp))
            for i in range(3):
                out.append((f"{pre}_line.{i}", getattr(self, f"{pre}_line"), i, "grid", grp))
        out.append(("appearance_basis_mat.weight", self.appearance_basis_mat, "weight", "matrix", "net_app"))

        def seq(prefix, s, grp):
            for j, m in enumerate(s):
                if isinstance(m, nn.Linear):
                    out.append((f"{prefix}.{j}.weight", m, "weight", "matrix", grp))
                    out.append((f"{prefix}.{j}.bias", m, "bias", "vector", grp))
        seq("render_appearance_mlp.mlp", self.render_appearance_mlp.mlp, "net_app")

        def grids(pre, grp):
            for i in range(3):
                out.append((f"{pre}_plane.{i}", getattr(self, f"{pre}_plane"), i, "grid", grp))
            for i in range(3):
                out.append((f"{pre}_line.{i}", getattr(self, f"{pre}_line"), i, "grid", grp))
        # the semantic head is its own optimizer range(s): while it has no gradient source (epoch < late_semantic_optimization)
        # the reference's Adam skips it (grad is None), so its step -- and its step COUNT -- must be skippable too
        if self.semantic_plane is not None:          # head on a VM grid: tables at the grid rate, basis matrix + MLP at the net rate
            grids("semantic", "grid_sem")
            out.append(("semantic_basis_mat.weight", self.semantic_basis_mat, "weight", "matrix", "net_sem"))
        seq("render_semantic_mlp.mlp", self.render_semantic_mlp.mlp, "net_sem")
        if self.instance_plane is not None:
            grids("instance", "grid_inst")
            # its own group: "inst_fast" must mirror "inst_slow" element for element (the EMA is one axpy over the two ranges)
            out.append(("instance_basis_mat.weight", self.instance_basis_mat, "weight", "matrix", "inst_basis"))
        if self.render_instance_mlp is not None:
            seq("render_instance_mlp.mlp", self.render_instance_mlp.mlp, "inst_fast")
            if self.slow_fast_mode:
                seq("render_instance_mlp.slow_mlp", self.render_instance_mlp.slow_mlp, "inst_slow")
        return out

    def pack(self, device=None):
        """(Re)build the arenas from the current parameter values and re-point every nn.Parameter at its arena view.
        Called at construction and after any operation that replaces parameters (upsample / shrink / .to())."""
        lay = self._layout()
        device = device or self.param_flat.device
        slots = []
        for name, owner, key, kind, grp in lay:
            p = owner[key] if isinstance(key, int) else getattr(owner, key)
            # the first appearance layer (150 inputs) gets a 160-float row pitch: a whole number of 8-chunk swizzle groups for the
            # persistent 128-wide layer kernel (csrc/layer_n128.hip); pad columns stay zero
            slots.append(Slot(name, p.shape, kind, grp, pitch_align=32 if name == "render_appearance_mlp.mlp.0.weight" else 4))
        arena = Arena(slots, device)
        pflat, gflat = arena.new_buffer(), arena.new_buffer()
        pv, gv = arena.views(pflat), arena.views(gflat)
        with torch.no_grad():
            for name, owner, key, kind, grp in lay:
                old = owner[key] if isinstance(key, int) else getattr(owner, key)
                pv[name].copy_(old.detach().to(device))
                new = nn.Parameter(pv[name], requires_grad=True)
                new.grad = gv[name]
                if isinstance(key, int):
                    owner[key] = new
                else:
                    setattr(owner, key, new)
        self.arena, self.param_flat, self.grad_flat = arena, pflat, gflat
        self._views, self._gviews = pv, gv
        return self

    def _apply(self, fn, recurse=True):
        # .to()/.cuda() re-create parameters; keep the arena invariant by re-packing on the new device
        r = super()._apply(fn, recurse)
        if getattr(self, "arena", None) is not None:
            dev = self.density_plane[0].device
            if dev != self.param_flat.device or self.density_plane[0].data_ptr() != self._views["density_plane.0"].data_ptr():
                self.pack(dev)
        return r

    def named_views(self):
        return self._views

    def xcd_workspace_for(self, prefix):
        """Make sure the accumulation copies of a table group exist (allocation + zero fill happen on the current
        stream, before any side-stream branch uses them)."""
        a, b = self.arena.range_of({"density": "grid_density", "appearance": "grid_app", "semantic": "grid_sem", "instance": "grid_inst"}[prefix])
        return self.xcd_workspace(prefix, 8 * (b - a))

    def xcd_workspace(self, key, numel):
        """Persistent zero-initialised scratch for the per-XCD gradient accumulation copies (engine.vm_grad_struct);
        clift_xcd_reduce leaves it zeroed again."""
        ws = self.__dict__.setdefault("_xcd_ws", {})
        t = ws.get(key)
        if t is None or t.numel() != numel or t.device != self.param_flat.device:
            t = torch.zeros(numel, dtype=torch.float32, device=self.param_flat.device)
            ws[key] = t
        return t

    def named_grad_views(self):
        return self._gviews

    def zero_grad_arena(self, *groups):
        if groups:
            a, b = self.arena.range_of(*groups)
            self.grad_flat[a:b].zero_()
        else:
            self.grad_flat.zero_()

    # ------------------------------------------------------------------ reference API: point-wise evaluation
    def compute_density_without_activation(self, xyz_sampled):
        from .engine import density_points
        return density_points(self, xyz_sampled, activation=False)

    def compute_density(self, xyz_sampled):
        from .engine import density_points
        return density_points(self, xyz_sampled, activation=True)

    def compute_appearance_feature(self, xyz_sampled):
        from .engine import appearance_feature_points
        return appearance_feature_points(self, xyz_sampled)

    def compute_semantic_feature(self, xyz_sampled):
        if self.use_semantic_mlp:
            return xyz_sampled      # tensoRF.py:142-144
        from .engine import grid_feature_points
        return grid_feature_points(self, "semantic", xyz_sampled)

    def compute_instance_feature(self, xyz_sampled):
        if self.use_instance_mlp:
            return xyz_sampled      # tensoRF.py:152-154
        from .engine import grid_feature_points
        return grid_feature_points(self, "instance", xyz_sampled)

    # ------------------------------------------------------------------ reference API: grid surgery (tensoRF.py:158-197)
    def _grid_lists(self):
        out = [(self.density_plane, self.density_line), (self.appearance_plane, self.appearance_line)]
        if self.semantic_plane is not None:
            out.append((self.semantic_plane, self.semantic_line))
        if self.instance_plane is not None:
            out.append((self.instance_plane, self.instance_line))
        return tuple(out)

    @torch.no_grad()
    def shrink(self, t_l, b_r):
        """Crop every plane/line to the voxel range [t_l, b_r) per axis (tensoRF.py:158-177), then re-pack the arena."""
        t_l = [int(x) for x in t_l]
        b_r = [int(x) for x in b_r]
        for planes, lines in self._grid_lists():
            for i in range(3):
                v = VECTOR_MODE[i]
                a, b = MATRIX_MODE[i]
                lines[i] = nn.Parameter(lines[i].data[..., t_l[v]:b_r[v], :].clone())
                planes[i] = nn.Parameter(planes[i].data[..., t_l[b]:b_r[b], t_l[a]:b_r[a]].clone())
        self.pack()

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """Bilinear (align_corners=True) resize of every plane/line to res_target = (Rx,Ry,Rz) (tensoRF.py:179-197).
        Runs 3-4 times per training: done with torch's interpolate on the device, then the arena is re-packed (which
        also makes the new tensors the ones the optimizer and the gradient all-reduce see)."""
        import torch.nn.functional as F
        res_target = [int(x) for x in res_target]
        for planes, lines in self._grid_lists():
            for i in range(3):
                v = VECTOR_MODE[i]
                a, b = MATRIX_MODE[i]
                src_p = planes[i].data.contiguous(memory_format=torch.contiguous_format)
                src_l = lines[i].data.contiguous(memory_format=torch.contiguous_format)
                planes[i] = nn.Parameter(F.interpolate(src_p, size=(res_target[b], res_target[a]), mode='bilinear', align_corners=True))
                lines[i] = nn.Parameter(F.interpolate(src_l, size=(res_target[v], 1), mode='bilinear', align_corners=True))
        self.pack()

    # ------------------------------------------------------------------ reference API: optimizer groups
    def get_optimizable_parameters(self, lr_grid, lr_net, weight_decay=0):
        """tensoRF.py:199-213."""
        g = [{'params': self.density_line, 'lr': lr_grid, 'weight_decay': weight_decay},
             {'params': self.appearance_line, 'lr': lr_grid},
             {'params': self.density_plane, 'lr': lr_grid, 'weight_decay': weight_decay},
             {'params': self.appearance_plane, 'lr': lr_grid},
             {'params': self.appearance_basis_mat.parameters(), 'lr': lr_net},
             {'params': self.render_appearance_mlp.parameters(), 'lr': lr_net}]
        if self.semantic_plane is not None:
            g += [{'params': self.semantic_plane, 'lr': lr_grid}, {'params': self.semantic_line, 'lr': lr_grid},
                  {'params': self.semantic_basis_mat.parameters(), 'lr': lr_net}]
        return g + [{'params': self.render_semantic_mlp.parameters(), 'lr': lr_net}]

    def get_optimizable_instance_parameters(self, lr_grid, lr_net, using_DINO=False):
        """tensoRF.py:229-246: (grid head: its planes / lines at the grid rate, the basis matrix,) the fast MLP; the slow MLP only when not DINO-style."""
        g = []
        if self.instance_plane is not None:
            g += [{'params': self.instance_plane, 'lr': lr_grid}, {'params': self.instance_line, 'lr': lr_grid},
                  {'params': self.instance_basis_mat.parameters(), 'lr': lr_net}]
        g.append({'params': self.render_instance_mlp.mlp.parameters(), 'lr': lr_net})
        if self.slow_fast_mode and not using_DINO:
            g.append({'params': self.render_instance_mlp.slow_mlp.parameters(), 'lr': lr_net})
        return g

    # ------------------------------------------------------------------ reference API: TV regulariser
    def total_tv_loss(self, regularizer=None, config=None, current_epoch=0, accumulate_grad=True, scale=1.0):
        """tensoRF.py:248-290: density and appearance PLANES x 1e-2, weighted by config.lambda_tv_density / lambda_tv_appearance; with a head on
        its own VM grid, that grid's planes x 1e-2 + LINES x 1e-3 weighted by lambda_tv_semantics / lambda_tv_instances, from
        late_semantic_optimization / instance_optimization_epoch on.  One streaming HIP launch per eight tables computes the value and
        (accumulate_grad) adds ``scale * d loss`` straight into the gradient arena -- except for the INSTANCE grid, whose term only enters the
        value: in the reference that gradient lands on parameters of the instance optimizer, which clears them (T:211) before its own
        backward, so it is never applied.  Returns a 0-dim device tensor (detached)."""
        g = lambda k, d: float(getattr(config, k, d)) if config is not None else d
        sem_on = config is None or current_epoch >= getattr(config, "late_semantic_optimization", 0)
        inst_on = config is None or current_epoch >= getattr(config, "instance_optimization_epoch", 0)
        terms = [(f"{pre}_plane.{i}", lam * 1e-2, True) for pre, lam in (("density", g("lambda_tv_density", 0.1)), ("appearance", g("lambda_tv_appearance", 0.01)))
                 for i in range(3)]
        if self.semantic_plane is not None and sem_on:
            lam = g("lambda_tv_semantics", 0.02)
            terms += [(f"semantic_plane.{i}", lam * 1e-2, True) for i in range(3)] + [(f"semantic_line.{i}", lam * 1e-3, True) for i in range(3)]
        if self.instance_plane is not None and inst_on:
            lam = g("lambda_tv_instances", 0.02)
            terms += [(f"instance_plane.{i}", lam * 1e-2, False) for i in range(3)] + [(f"instance_line.{i}", lam * 1e-3, False) for i in range(3)]
        out = torch.zeros(1, dtype=torch.float32, device=self.param_flat.device)
        for c0 in range(0, len(terms), 8):
            ts = _lib.TVSet()
            for n, (name, wgt, with_grad) in enumerate(terms[c0:c0 + 8]):
                p = self._views[name]
                gr = self._gviews[name] if (accumulate_grad and with_grad) else None
                _, c, h, w = p.shape
                ts.plane[n], ts.grad[n] = p.data_ptr(), (gr.data_ptr() if gr is not None else None)
                ts.H[n], ts.W[n], ts.C[n], ts.weight[n] = h, w, c, float(wgt * scale)
            ts.n = len(terms[c0:c0 + 8])
            _lib.call("clift_tv_fwd_bwd_multi", C.byref(ts), _lib.ptr(out), _lib.stream())     # (all six planes of the MLP-heads field in one launch)
        return out[0] / scale if scale != 1.0 else out[0]

    # ------------------------------------------------------------------ checkpoint helpers
    def export_state_dict(self):
        """state_dict with plain contiguous tensors in the reference's layout (what a Lightning .ckpt holds)."""
        return {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in self.state_dict().items()}
