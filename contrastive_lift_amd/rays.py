"""Ray generation -- mirror of reference util/ray.py:8-12,25-31,46-54,81-99.

``generate_ray_table`` is the fused device path (clift_gen_rays): pixel grid -> directions -> world rays ->
unit-sphere far bound -> the (H*W, 8) record [o, d, near, far] the renderer consumes, in one kernel.  The four
reference-named functions are provided on top of it for call-site compatibility.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def generate_ray_table(height, width, intrinsics, cam2world, near=0.01, device=None, check=True):
    """(H*W, 8) float32 rays on ``device``; raises AssertionError like the reference (util/ray.py:96-98) if a ray
    misses the unit sphere (camera outside it)."""
    device = torch.device(device or "cuda")
    K = np.ascontiguousarray(np.asarray(torch.as_tensor(intrinsics).cpu(), dtype=np.float32)[:3, :3])
    M = np.ascontiguousarray(np.asarray(torch.as_tensor(cam2world).cpu(), dtype=np.float32)[:4, :4])
    rays = torch.empty((height * width, 8), dtype=torch.float32, device=device)
    bad = torch.zeros((1,), dtype=torch.int32, device=device)
    _lib.call("clift_gen_rays", int(height), int(width), K.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p),
              float(near), _lib.ptr(rays), _lib.ptr(bad), _lib.stream())
    if check and int(bad.item()) != 0:
        raise AssertionError("Not all your cameras are bounded by the unit sphere; please make sure the cameras are "
                             "normalized properly!")
    return rays


def create_grid(height, width):
    xs = torch.arange(width, dtype=torch.float32)[None, :].expand(height, width)
    ys = torch.arange(height, dtype=torch.float32)[:, None].expand(height, width)
    return xs, ys


def get_ray_directions_with_intrinsics(height, width, intrinsics):
    i, j = create_grid(height, width)
    fx, fy, cx, cy = intrinsics[0, 0], intrinsics[1, 1], intrinsics[0, 2], intrinsics[1, 2]
    return torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)


def get_rays(directions, cam2world):
    d = directions @ cam2world[:3, :3].T
    d = d / torch.norm(d, dim=-1, keepdim=True)
    o = cam2world[:3, 3].expand(d.shape)
    return o.reshape(-1, 3), d.reshape(-1, 3)


def rays_intersect_sphere(rays_o, rays_d, r=1):
    od = torch.sum(rays_o * rays_d, 1)
    dd = torch.sum(rays_d ** 2, 1)
    oo = torch.sum(rays_o ** 2, 1)
    disc = od ** 2 + (r ** 2 - oo) * dd
    assert torch.all(disc >= 0), \
        "Not all your cameras are bounded by the unit sphere; please make sure the cameras are normalized properly!"
    return (torch.sqrt(disc) - od) / dd
