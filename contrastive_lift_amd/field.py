"""TensorVMSplit -- drop-in mirror of the reference radiance field (model/radiance_field/tensoRF.py:32-315).

Same constructor signature, attribute names, ``state_dict`` keys/shapes and optimizer param groups as the
reference; the arithmetic runs in libclift.so (HIP, gfx950).  Storage differs from the reference on purpose:
every trainable tensor is a view into one flat arena (arena.py) -- VM tables channels-last, weight matrices
with a 16-byte row pitch -- while keeping the reference's logical shapes.

Supported configuration: ``use_semantic_mlp=True`` and ``use_instance_mlp=True`` (the template default and
every shipped contrastive-lift config, config/template/panopli_paper.yaml:36-37).  Grid semantic/instance heads
and distilled-feature grids (reference tensoRF.py:70-83,91-94) raise NotImplementedError (next).
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
        from .engine import xyz_mlp_points
        out = xyz_mlp_points(self.mlp, feat_xyz)
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
        from .engine import xyz_mlp_points
        out = xyz_mlp_points(self.mlp, feat_xyz)
        if self.slow_fast_mode:
            out = torch.cat([out, xyz_mlp_points(self.slow_mlp, feat_xyz)], -1)
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
        if not (use_semantic_mlp and use_instance_mlp):
            raise NotImplementedError("clift: grid semantic/instance heads are not built yet (SURVEY 8f 'next'); "
                                      "use use_semantic_mlp=True, use_instance_mlp=True (the template default)")
        if use_distilled_features_semantic or use_distilled_features_instance or use_proj or use_feature_reg:
            raise NotImplementedError("clift: distilled-feature grids / projection head / feature regulariser are off "
                                      "in every shipped contrastive-lift config and are not built")
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
            self.render_instance_mlp = MLPRenderInstanceFeature(3, ins_out, num_mlp_layers=4, dim_mlp=dim_mlp_instance,
                                                                slow_fast_mode=slow_fast_mode)
        self.render_semantic_mlp = MLPRenderSemanticFeature(3, num_semantic_classes, softmax=softmax)
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
        # the semantic MLP is its own optimizer range: while it has no gradient source (epoch < late_semantic_optimization)
        # the reference's Adam skips it (grad is None), so its step -- and its step COUNT -- must be skippable too
        seq("render_semantic_mlp.mlp", self.render_semantic_mlp.mlp, "net_sem")
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
        a, b = self.arena.range_of({"density": "grid_density", "appearance": "grid_app"}[prefix])
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
        return xyz_sampled          # use_semantic_mlp (tensoRF.py:142-144)

    def compute_instance_feature(self, xyz_sampled):
        return xyz_sampled          # use_instance_mlp (tensoRF.py:152-154)

    # ------------------------------------------------------------------ reference API: grid surgery (tensoRF.py:158-197)
    def _grid_lists(self):
        return ((self.density_plane, self.density_line), (self.appearance_plane, self.appearance_line))

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
        """tensoRF.py:199-213 (MLP-heads configuration)."""
        return [{'params': self.density_line, 'lr': lr_grid, 'weight_decay': weight_decay},
                {'params': self.appearance_line, 'lr': lr_grid},
                {'params': self.density_plane, 'lr': lr_grid, 'weight_decay': weight_decay},
                {'params': self.appearance_plane, 'lr': lr_grid},
                {'params': self.appearance_basis_mat.parameters(), 'lr': lr_net},
                {'params': self.render_appearance_mlp.parameters(), 'lr': lr_net},
                {'params': self.render_semantic_mlp.parameters(), 'lr': lr_net}]

    def get_optimizable_instance_parameters(self, lr_grid, lr_net, using_DINO=False):
        """tensoRF.py:229-246: fast MLP always; slow MLP only when not DINO-style."""
        g = [{'params': self.render_instance_mlp.mlp.parameters(), 'lr': lr_net}]
        if self.slow_fast_mode and not using_DINO:
            g.append({'params': self.render_instance_mlp.slow_mlp.parameters(), 'lr': lr_net})
        return g

    # ------------------------------------------------------------------ reference API: TV regulariser
    def total_tv_loss(self, regularizer=None, config=None, current_epoch=0, accumulate_grad=True, scale=1.0):
        """tensoRF.py:248-258,281-290 for the MLP-heads configuration: density and appearance PLANES, x1e-2 each,
        weighted by config.lambda_tv_density / lambda_tv_appearance.  One streaming HIP pass per plane computes the
        value and (accumulate_grad) adds ``scale * d loss`` straight into the gradient arena.
        Returns a 0-dim device tensor (detached)."""
        lam_d = float(getattr(config, "lambda_tv_density", 0.1)) if config is not None else 0.1
        lam_a = float(getattr(config, "lambda_tv_appearance", 0.01)) if config is not None else 0.01
        out = torch.zeros(1, dtype=torch.float32, device=self.param_flat.device)
        ts = _lib.TVSet()
        n = 0
        for pre, lam in (("density", lam_d), ("appearance", lam_a)):
            for i in range(3):
                p = self._views[f"{pre}_plane.{i}"]
                g = self._gviews[f"{pre}_plane.{i}"] if accumulate_grad else None
                _, c, h, w = p.shape
                ts.plane[n], ts.grad[n] = p.data_ptr(), (g.data_ptr() if g is not None else None)
                ts.H[n], ts.W[n], ts.C[n], ts.weight[n] = h, w, c, float(lam * 1e-2 * scale)
                n += 1
        ts.n = n
        _lib.call("clift_tv_fwd_bwd_multi", C.byref(ts), _lib.ptr(out), _lib.stream())     # all six planes in one launch
        return out[0] / scale if scale != 1.0 else out[0]

    # ------------------------------------------------------------------ checkpoint helpers
    def export_state_dict(self):
        """state_dict with plain contiguous tensors in the reference's layout (what a Lightning .ckpt holds)."""
        return {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in self.state_dict().items()}
