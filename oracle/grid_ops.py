"""Oracle: alpha-mask shrink and grid upsampling -- reference model/renderer/panopli_tensoRF_renderer.py:668-761
(update_bbox_aabb_and_shrink, get_dense_alpha, compute_alpha, get_target_resolution), model/radiance_field/tensoRF.py:158-197
(shrink, upsample_volume_grid) and the voxel schedule of trainer/train_panopli_tensorf.py:451.

Test infrastructure only (see oracle/__init__.py).  Pinned by tests/golden/g10_grid_ops.npz (each operation) and g21_epoch_boundary.npz
(composed, inside the reference trainer's epoch hook).  Functions mutate the parameter dict ``P`` (new tensors under the same keys,
like the reference's new nn.Parameters) and the RenderCfg.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import field as fld
from .params import MATRIX_MODE, VECTOR_MODE

GRID_PREFIXES = ("density", "appearance", "semantic", "instance")


def dense_alpha(P, cfg):
    """R:717-729, 750-754: alpha = 1 - exp(-sigma * step_size) on the (Rx, Ry, Rz) lattice of the current box."""
    g = [int(x) for x in cfg.grid_dim]
    samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, g[0]), torch.linspace(0, 1, g[1]), torch.linspace(0, 1, g[2]), indexing="ij"), -1)
    dense_xyz = cfg.aabb[0] * (1 - samples) + cfg.aabb[1] * samples
    xn = (dense_xyz.reshape(-1, 3) - cfg.aabb[0]) * cfg.inv_extent2 - 1                     # R:633-634
    sigma = fld.density(P, xn, shift=cfg.density_shift).reshape(g)
    return 1 - torch.exp(-sigma * cfg.step_size), dense_xyz


@torch.no_grad()
def shrink(P, t_l, b_r):
    """F:158-177: crop every line to [t_l, b_r) of its axis, every plane to the two axes it spans."""
    for pre in GRID_PREFIXES:
        if f"{pre}_plane.0" not in P:
            continue
        for i in range(3):
            v = VECTOR_MODE[i]
            a, b = MATRIX_MODE[i]
            P[f"{pre}_line.{i}"] = P[f"{pre}_line.{i}"].detach()[..., t_l[v]:b_r[v], :].clone().requires_grad_(True)
            P[f"{pre}_plane.{i}"] = P[f"{pre}_plane.{i}"].detach()[..., t_l[b]:b_r[b], t_l[a]:b_r[a]].clone().requires_grad_(True)


@torch.no_grad()
def update_bbox_aabb_and_shrink(P, cfg, alpha_mask_threshold=0.0075, fractional_lenience=1.0):
    """R:668-715.  Returns True when the box changed."""
    alpha, dense_xyz = dense_alpha(P, cfg)
    g = torch.tensor([int(x) for x in cfg.grid_dim])
    dense_xyz = dense_xyz.transpose(0, 2).contiguous()
    alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
    alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(g.tolist()[::-1])
    occ = alpha >= alpha_mask_threshold
    valid = dense_xyz[occ]
    if valid.shape[0] == 0:
        return False
    xyz_min, xyz_max = valid.amin(0), valid.amax(0)
    extent, position = xyz_max - xyz_min, (xyz_min + xyz_max) / 2
    xyz_min = torch.maximum(cfg.aabb[0], position - (extent * fractional_lenience) / 2)
    xyz_max = torch.minimum(cfg.aabb[1], position + (extent * fractional_lenience) / 2)
    t_l, b_r = (xyz_min - cfg.aabb[0]) / cfg.units, (xyz_max - cfg.aabb[0]) / cfg.units
    t_l, b_r = torch.round(torch.round(t_l)).long(), torch.round(b_r).long() + 1
    b_r = torch.stack([b_r, g]).amin(0)
    new_size = b_r - t_l
    if not bool((new_size > 0).all()):
        return False
    shrink(P, t_l.tolist(), b_r.tolist())
    cfg.aabb = torch.stack((xyz_min, xyz_max))
    cfg.grid_dim = tuple(int(x) for x in new_size.tolist())
    cfg.refresh()
    return True


@torch.no_grad()
def upsample_volume_grid(P, res_target):
    """F:179-197: bilinear, align_corners=True."""
    for pre in GRID_PREFIXES:
        if f"{pre}_plane.0" not in P:
            continue
        for i in range(3):
            v = VECTOR_MODE[i]
            a, b = MATRIX_MODE[i]
            P[f"{pre}_plane.{i}"] = F.interpolate(P[f"{pre}_plane.{i}"].detach(), size=(res_target[b], res_target[a]), mode="bilinear",
                                                  align_corners=True).requires_grad_(True)
            P[f"{pre}_line.{i}"] = F.interpolate(P[f"{pre}_line.{i}"].detach(), size=(res_target[v], 1), mode="bilinear",
                                                 align_corners=True).requires_grad_(True)


def target_resolution(cfg, n_voxels):
    """R:756-761."""
    ext = cfg.aabb[1] - cfg.aabb[0]
    voxel = (ext.prod() / n_voxels).pow(1 / 3)
    return tuple(max(int(x), 1) for x in (ext / voxel).long().tolist())


def voxel_schedule(min_grid_dim, max_grid_dim, n_upscales):
    """T:451: log-spaced voxel counts, one per grid_upscale epoch."""
    return torch.round(torch.exp(torch.linspace(np.log(min_grid_dim ** 3), np.log(max_grid_dim ** 3), n_upscales + 1))).long().tolist()[1:]
