"""Hydra-less config objects for the sampler path.

The reference drives everything through an OmegaConf ``DictConfig`` and relies on
four access idioms: attribute access, ``"key" in cfg.keys()``,
``cfg.get(k, default)`` and ``hasattr(cfg, "loss_1")`` (reference ``utils/losses.py:22,31,46,74``,
``testing/tester.py:105,172``).  ``AttrDict`` supports all four; ``load_yaml`` reads a
``conf/``-style YAML file (PyYAML parses ``1e-4`` as a string, so numeric-looking strings
are coerced the way OmegaConf would).
"""
from __future__ import annotations

import os
import re

import yaml

_NUM_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")


class AttrDict(dict):
    """dict with attribute access; missing attributes raise AttributeError (so hasattr works)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name) from None

    def copy(self):
        return to_attrdict({k: v for k, v in self.items()})


def _coerce(v):
    if isinstance(v, str):
        s = v.strip()
        if _NUM_RE.match(s):
            f = float(s)
            if re.match(r"^[+-]?\d+$", s):
                return int(s)
            return f
        if s == "None":
            return None
    return v


def to_attrdict(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attrdict(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_attrdict(v) for v in obj]
    return _coerce(obj)


def load_yaml(path):
    with open(path, "r") as f:
        return to_attrdict(yaml.safe_load(f))


def merge(base, override):
    """Recursive dict merge (override wins); returns a new AttrDict."""
    out = to_attrdict(dict(base))
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = to_attrdict(v)
    return out


def set_by_path(cfg, dotted, value):
    """``tester.sampling_params.T=50`` style override."""
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            node[k] = AttrDict()
        node = node[k]
    node[keys[-1]] = to_attrdict(value)


CONF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")


def compose(tester="blind_dereverberation_BUDDy", network="ncsnpp", diff_params="edm_VCTK",
            exp="VCTK_16k_4s_time", overrides=(), conf_dir=None):
    """Compose the config tree the way ``conf/conf_VCTK.yaml`` + Hydra groups do in the reference
    (reference ``conf/conf_VCTK.yaml:1-7``): ``args.network``, ``args.diff_params``, ``args.tester``,
    ``args.exp``.  ``overrides`` is an iterable of ``"a.b.c=value"`` strings."""
    conf_dir = conf_dir or CONF_DIR
    args = AttrDict()
    args.network = load_yaml(os.path.join(conf_dir, "network", network + ".yaml"))
    args.diff_params = load_yaml(os.path.join(conf_dir, "diff_params", diff_params + ".yaml"))
    args.tester = load_yaml(os.path.join(conf_dir, "tester", tester + ".yaml"))
    args.exp = load_yaml(os.path.join(conf_dir, "exp", exp + ".yaml"))
    args.model_dir = "experiments"
    for ov in overrides:
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        set_by_path(args, k, yaml.safe_load(v))
    return args
