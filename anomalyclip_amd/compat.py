"""Alias installer: makes the reference's own Hydra `_target_` strings resolve to the libacx-backed mirrors, so the
reference's YAML (configs/model/anomaly_clip_*.yaml:1,10,16,46) runs unedited.

    import anomalyclip_amd.compat as compat
    compat.install()          # before hydra.utils.instantiate(cfg.model)

After install(), `importlib.import_module("src.models.components.anomaly_clip").AnomalyCLIP` is
anomalyclip_amd.components.anomaly_clip.AnomalyCLIP, and likewise for ComputeLoss, WarmupCosineAnnealingLR and
AnomalyCLIPModule.  Only these four hot-path modules are aliased; everything else under `src.` (datamodule, utils,
train.py / eval.py) stays the reference's when the reference checkout is on sys.path, and parent packages are created
as empty namespace stand-ins only when it is not.  `torch.optim.AdamW` is left alone: it works on the mirrors'
parameters as it is (set `optimizer._target_: anomalyclip_amd.optim.AcxAdamW` for the fused libacx update)."""
from __future__ import annotations

import importlib
import sys
import types

ALIASES = {
    "src.models.components.anomaly_clip": "anomalyclip_amd.components.anomaly_clip",
    "src.models.components.loss": "anomalyclip_amd.components.loss",
    "src.models.components.scheduler": "anomalyclip_amd.components.scheduler",
    "src.models.anomaly_clip_module": "anomalyclip_amd.anomaly_clip_module",
}
_installed: dict = {}


def _parent_package(name: str) -> types.ModuleType:
    """the real package when the reference checkout is importable, otherwise an empty stand-in package."""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:  # noqa: BLE001 - absent, or its __init__ needs packages this environment lacks
        mod = types.ModuleType(name)
        mod.__path__ = []                     # a package: submodule imports consult sys.modules first
        mod.__acx_alias_parent__ = True
        sys.modules[name] = mod
        _installed[name] = None
        return mod


def install() -> dict:
    """Idempotent.  Returns {reference module name: mirror module}."""
    out = {}
    for ref_name, mirror_name in ALIASES.items():
        mirror = importlib.import_module(mirror_name)
        parts = ref_name.split(".")
        for i in range(1, len(parts)):
            parent = _parent_package(".".join(parts[:i]))
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], parent)
        if ref_name not in _installed:
            _installed[ref_name] = sys.modules.get(ref_name)
        sys.modules[ref_name] = mirror
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mirror)
        out[ref_name] = mirror
    return out


def uninstall() -> None:
    """Restore sys.modules as it was before install()."""
    for name, prev in list(_installed.items()):
        if prev is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = prev
    _installed.clear()


def resolve(target: str):
    """`_target_` string -> object, the way hydra.utils.get_class does it (import the module path, getattr the rest)."""
    mod, _, attr = target.rpartition(".")
    return getattr(importlib.import_module(mod), attr)
