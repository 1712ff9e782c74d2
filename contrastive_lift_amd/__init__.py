"""contrastive_lift_amd -- MI355X-native (HIP, gfx950) rendering hot path of Contrastive-Lift.

Drop-in mirrors of the reference's objects for this path:
    TensorVMSplit      (reference model/radiance_field/tensoRF.py)
    TensoRFRenderer    (reference model/renderer/panopli_tensoRF_renderer.py)
    contrastive_loss, TVLoss, SCELoss, get_semantic_weights, slow_fast_loss, ema_update (reference model/loss/loss.py, trainer T:256-329)
    ray generation     (reference util/ray.py)
All arithmetic runs in libclift.so (include/clift.h); there is no CPU / PyTorch fallback.
"""
from ._lib import CliftError, build, load  # noqa: F401
from .field import TensorVMSplit  # noqa: F401
from .renderer import TensoRFRenderer  # noqa: F401
from .loss import (TVLoss, SCELoss, SoftTargetCrossEntropy, get_semantic_weights, contrastive_loss, slow_fast_loss,  # noqa: F401
                   ema_update)
from .rays import (create_grid, get_ray_directions_with_intrinsics, get_rays, rays_intersect_sphere,  # noqa: F401
                   generate_ray_table)
