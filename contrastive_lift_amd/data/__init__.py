from .mos import MOSScene  # noqa: F401
