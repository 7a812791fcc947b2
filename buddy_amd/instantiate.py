"""Minimal stand-in for ``hydra.utils.instantiate`` (Hydra is not installable offline): resolves the reference's
``_target_`` class paths (the plug-point API) onto this package's implementations."""
from __future__ import annotations

import importlib

_ALIASES = {
    "networks.ncsnpp.NCSNppTime": "buddy_amd.networks.ncsnpp.NCSNppTime",
    "diff_params.edm.EDM": "buddy_amd.diff_params.edm.EDM",
    "testing.EulerHeunSampler.EulerHeunSampler": "buddy_amd.testing.EulerHeunSampler.EulerHeunSampler",
    "testing.EulerHeunSamplerDPS.EulerHeunSamplerDPS": "buddy_amd.testing.EulerHeunSamplerDPS.EulerHeunSamplerDPS",
    "testing.tester.Tester": "buddy_amd.testing.tester.Tester",
    "datasets.vctk.VCTKTestPaired": "buddy_amd.datasets.vctk.VCTKTestPaired",
}


def resolve(target):
    path = _ALIASES.get(target, target)
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, *args, **kwargs):
    kw = {k: v for k, v in cfg.items() if k != "_target_"}
    kw.update(kwargs)
    return resolve(cfg["_target_"])(*args, **kw)
