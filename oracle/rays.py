"""Oracle: ray generation.  Test infrastructure only (see oracle/__init__.py)."""
import torch


def pixel_grid(height, width):
    """Pixel index images: ``i`` = column index 0..W-1, ``j`` = row index 0..H-1, both (H, W).
    No half-pixel offset.  Follows reference util/ray.py:8-12."""
    i = torch.arange(width, dtype=torch.float32)[None, :].expand(height, width)
    j = torch.arange(height, dtype=torch.float32)[:, None].expand(height, width)
    return i, j


def camera_dirs(height, width, K):
    """Camera-space directions ((i-cx)/fx, (j-cy)/fy, 1), +z forward.  util/ray.py:25-31."""
    i, j = pixel_grid(height, width)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    return torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)


def world_rays(dirs, cam2world):
    """Rotate by R, L2-normalise, origin = translation; flattened row-major.  util/ray.py:46-54."""
    R = cam2world[:3, :3].to(torch.float32)
    d = dirs @ R.T
    d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    o = cam2world[:3, 3].to(torch.float32).expand(d.shape)
    return o.reshape(-1, 3), d.reshape(-1, 3)


def sphere_far(o, d, r=1.0):
    """Forward intersection of o + t d with the sphere |x| = r.  util/ray.py:81-99 (asserts D>=0)."""
    od = (o * d).sum(1)
    dd = (d * d).sum(1)
    oo = (o * o).sum(1)
    disc = od * od + (r * r - oo) * dd
    if not bool((disc >= 0).all()):
        raise AssertionError("Not all your cameras are bounded by the unit sphere; "
                             "please make sure the cameras are normalized properly!")
    return (torch.sqrt(disc) - od) / dd


def ray_table(height, width, K, cam2world, near=0.01):
    """Ray record [o(3), d(3), near, far] (P, 8); reference dataset/many_object_scenes.py:191-199."""
    o, d = world_rays(camera_dirs(height, width, K), cam2world)
    far = sphere_far(o, d, 1.0)
    return torch.cat([o, d, torch.full_like(far[:, None], near), far[:, None]], 1)
