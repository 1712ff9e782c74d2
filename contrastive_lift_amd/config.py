"""Config tree loader for the reference's Hydra layout without Hydra/OmegaConf (absent from this image).

Layout understood (reference config/): ``config.yaml`` with ``defaults: [{template: <name>}]``,
``template/<name>.yaml`` holding every knob, ``experiment/<name>.yaml`` overlays marked ``# @package _global_`` whose
``template:`` mapping overrides template keys.  ``load_config`` returns the *template* node -- exactly what the
reference's ``main`` hands to its trainer (trainer/train_panopli_tensorf.py:479-480) -- as an attribute dict.
Command-line overrides follow Hydra's syntax: ``+experiment=contrastive_lift_MOS template.lr=1e-3 dataset_root=...``.
"""
import os
import re

import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")


def _numbers(x):
    """PyYAML (YAML 1.1) reads ``5e-4`` / ``1e-8`` as strings; OmegaConf reads them as floats.  Match OmegaConf."""
    if isinstance(x, str) and _FLOAT.match(x):
        return float(x)
    if isinstance(x, dict):
        return {k: _numbers(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_numbers(v) for v in x]
    return x


def _read(path):
    with open(path) as f:
        return _numbers(yaml.safe_load(f) or {})


def _coerce(v):
    try:
        return _numbers(yaml.safe_load(v))
    except yaml.YAMLError:
        return v


def load_config(config_dir, experiment=None, overrides=()):
    root = _read(os.path.join(config_dir, "config.yaml"))
    template = "panopli_paper"
    for d in root.get("defaults", []):
        if isinstance(d, dict) and "template" in d:
            template = d["template"]
    pending = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if k == "experiment":
            experiment = v
        elif k == "template" and "." not in k:
            template = v
        else:
            pending.append((k[len("template."):] if k.startswith("template.") else k, _coerce(v)))
    cfg = AttrDict(_read(os.path.join(config_dir, "template", f"{template}.yaml")))
    cfg.pop("hydra", None)
    if experiment:
        exp = _read(os.path.join(config_dir, "experiment", f"{experiment}.yaml"))
        for d in exp.get("defaults", []) or []:
            if isinstance(d, dict):
                for dk, dv in d.items():
                    if dk.replace("override ", "").strip("/ ") == "template" and dv != template:
                        base = AttrDict(_read(os.path.join(config_dir, "template", f"{dv}.yaml")))
                        base.update(cfg)
                        cfg = base
        cfg.update(exp.get("template", {}) or {})
    for k, v in pending:
        cfg[k] = v
    return cfg


def save_config(cfg, path):
    """runs/<experiment>/config.yaml, the file inference/render_panopli.py loads (util/filesystem_logger.py:57, RP:445)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        yaml.safe_dump(dict(cfg), f, sort_keys=False)


def load_run_config(path):
    return AttrDict(_read(path))
