"""Minimal stand-in for the hydra/omegaconf layer of the reference (``@hydra.main(config_path="../bioscanclip/config",
config_name="global_config")``, scripts/train_cl.py:245): reads the same YAML files and ``key=value`` /
``model_config=<name>`` command-line overrides into an attribute tree with the ``hasattr`` semantics the reference
probes optional keys with (train_cl.py:129,155,160,184; simple_clip.py:138,163,175-176).  hydra/omegaconf are not
installed in this image; PyYAML is."""
import os

import yaml


class AttrDict(dict):
    """dict with attribute access; missing keys raise AttributeError so ``hasattr(cfg, 'x')`` works like OmegaConf."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(x):
    if isinstance(x, dict):
        return AttrDict({k: to_attr(v) for k, v in x.items()})
    if isinstance(x, list):
        return [to_attr(v) for v in x]
    return x


def _parse_scalar(s):
    try:
        return yaml.safe_load(s)
    except Exception:
        return s


def _set_path(cfg, dotted, value):
    node = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            node[p] = AttrDict()
        node = node[p]
    node[parts[-1]] = value


def load_config(config_dir, overrides=(), config_name="global_config"):
    """``overrides``: hydra-style strings.  ``model_config=<name>`` selects ``<config_dir>/model_config/<name>.yaml``
    (the reference's default ``mlp_ssl`` does not exist, so the override is mandatory there too: SURVEY App. B-5)."""
    path = os.path.join(config_dir, config_name + ".yaml")
    cfg = AttrDict()
    if os.path.exists(path):
        with open(path) as f:
            cfg = to_attr(yaml.safe_load(f) or {})
    defaults = cfg.pop("defaults", None)
    model_cfg_name = None
    if isinstance(defaults, list):
        for d in defaults:
            if isinstance(d, dict) and "model_config" in d:
                model_cfg_name = d["model_config"]
    rest = []
    for ov in overrides:
        ov = ov.strip().strip("'").strip('"')
        if ov.startswith("model_config="):
            model_cfg_name = ov.split("=", 1)[1]
        else:
            rest.append(ov)
    if model_cfg_name is not None:
        mpath = os.path.join(config_dir, "model_config", str(model_cfg_name) + ".yaml")
        if not os.path.exists(mpath):
            raise FileNotFoundError(f"model_config '{model_cfg_name}' not found at {mpath}")
        with open(mpath) as f:
            cfg["model_config"] = to_attr(yaml.safe_load(f) or {})
    for ov in rest:
        k, v = ov.split("=", 1)
        _set_path(cfg, k.lstrip("+"), _parse_scalar(v))
    return cfg
