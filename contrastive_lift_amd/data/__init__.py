from .mos import MOSScene  # noqa: F401
from .panopli import PanopLiScene  # noqa: F401


def get_scene(config, split, device):
    """dataset/__init__.py:9-41: the reader for ``config.dataset_class`` with the reference's default label directories."""
    if config.dataset_class == "mos":
        return MOSScene(config.dataset_root, split, config.image_dim, config.max_depth, subsample_frames=config.subsample_frames, device=device)
    if config.dataset_class == "panopli":
        return PanopLiScene(config.dataset_root, split, config.image_dim, config.max_depth, subsample_frames=config.subsample_frames, device=device)
    raise NotImplementedError(f"dataset_class {config.dataset_class!r}: expected 'mos' or 'panopli'")
