"""Import harness that makes the reference's hot-path modules importable in the DEVELOPMENT
container (where /root/reference exists).  Used only by tests/golden/make_golden.py.

Nothing here is shipped or used at run time; the GPU box has no /root/reference.

What is stubbed (modules absent from this image; none of them carries hot-path arithmetic):
  torchvision(+transforms)  -- only imported for CLIP's PIL preprocessing, never executed here
  ftfy                      -- fix_text = identity (class names are ASCII)
  dotmap.DotMap             -- attribute dict returning None for missing keys
  src.utils                 -- get_pylogger only
  axial_attention           -- oracle/axial_attention_restated.py (PARITY UNPINNED, see there)
`clip.load` is monkey-patched to return a seeded random-init CLIP of the requested geometry
(no network, no checkpoints in this image).
"""
import importlib
import logging
import os
import sys
import types

REF_ROOT = os.environ.get("ACX_REFERENCE_ROOT", "/root/reference")
AXIAL_SOURCE = "not installed yet"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _DotMap(dict):
    def __init__(self, **kw):
        super().__init__(**kw)

    def __getattr__(self, k):
        return self.get(k, None)


def install():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)

    # torchvision stub
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for name in ("CenterCrop", "Compose", "Normalize", "Resize", "ToTensor"):
        setattr(tvt, name, type(name, (), {"__init__": lambda self, *a, **k: None}))

    class _IM:
        BICUBIC = 3

    tvt.InterpolationMode = _IM
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)

    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    sys.modules.setdefault("ftfy", ftfy)

    dm = types.ModuleType("dotmap")
    dm.DotMap = _DotMap
    sys.modules.setdefault("dotmap", dm)

    # `src` package: real package dir, but src.utils replaced by a stub (the real one pulls
    # hydra / lightning).
    import src  # noqa: F401  (reference package, empty __init__)

    su = types.ModuleType("src.utils")
    su.get_pylogger = lambda name=__name__: logging.getLogger(name)
    sys.modules["src.utils"] = su
    sys.modules["src"].utils = su

    # a7: the REAL `axial_attention` package when it is importable (the day the wheel is reachable the fixtures pin
    # themselves: AXIAL_SOURCE is stamped into every fixture that runs the temporal model); the restatement otherwise
    global AXIAL_SOURCE
    try:
        if getattr(sys.modules.get("axial_attention"), "__acx_restated__", False):
            del sys.modules["axial_attention"]                   # a second install(): look for the real package again
        real = importlib.import_module("axial_attention")
        if getattr(real, "__acx_restated__", False):
            raise ImportError("only the restatement is on the path")
        ver = "unknown"
        try:
            from importlib import metadata
            ver = metadata.version("axial_attention")
        except Exception:  # noqa: BLE001
            pass
        AXIAL_SOURCE = f"pypi:axial_attention=={ver}"
    except ImportError:
        from oracle import axial_attention_restated as ax
        ax.__acx_restated__ = True
        sys.modules["axial_attention"] = ax
        AXIAL_SOURCE = "restated:oracle/axial_attention_restated.py (parity unpinned)"


def ref_modules():
    """Returns a namespace with the reference modules on the hot path."""
    install()
    ns = types.SimpleNamespace()
    ns.clip_model = importlib.import_module("src.models.components.clip.model")
    ns.clip = importlib.import_module("src.models.components.clip.clip")
    ns.selector_model = importlib.import_module("src.models.components.selector_model")
    ns.loss = importlib.import_module("src.models.components.loss")
    ns.classification_head = importlib.import_module("src.models.components.classification_head")
    ns.text_encoder = importlib.import_module("src.models.components.text_encoder")
    ns.temporal_model = importlib.import_module("src.models.components.temporal_model")
    ns.coop = importlib.import_module("src.models.components.coop")
    ns.anomaly_clip = importlib.import_module("src.models.components.anomaly_clip")
    ns.scheduler = importlib.import_module("src.models.components.scheduler")
    return ns


def patch_clip_load(ns, geometry, seed):
    """clip.load -> seeded random-init CLIP(**geometry) (reference class, reference init)."""
    import torch

    def _load(name, device="cpu", jit=False, download_root=None):
        torch.manual_seed(seed)
        m = ns.clip_model.CLIP(**geometry)
        return m.eval(), None

    ns.clip.load = _load
    # anomaly_clip.py did `from src.models.components.clip import clip` -> same module object
    return _load
