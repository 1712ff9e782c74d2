from .mos import MOSScene  # noqa: F401
from .panopli import PanopLiScene  # noqa: F401


def get_scene(config, split, device, image_dim=None):
    """dataset/__init__.py:9-41: the reader for ``config.dataset_class`` with the reference's default label directories.
    ``image_dim`` overrides config.image_dim (the segment dataset is always built at (128, 128), dataset/__init__.py:70,78)."""
    dim = image_dim if image_dim is not None else config.image_dim
    if config.dataset_class == "mos":
        return MOSScene(config.dataset_root, split, dim, config.max_depth, subsample_frames=config.subsample_frames, device=device)
    if config.dataset_class == "panopli":
        return PanopLiScene(config.dataset_root, split, dim, config.max_depth, subsample_frames=config.subsample_frames, device=device)
    raise NotImplementedError(f"dataset_class {config.dataset_class!r}: expected 'mos' or 'panopli'")
