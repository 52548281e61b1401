"""Checkpoint interop: load the `state_dict` of a reference Lightning `.ckpt`
(`configs/callbacks/default.yaml:8-14`; keys `net.<module path>`) and the `ncentroid.pt` side-car
(`anomaly_clip_module.py:140-171`) into the mirrors, and write checkpoints the reference can read back."""
from __future__ import annotations

from typing import Dict, Mapping, Tuple

import torch


def split_lightning_state_dict(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`net.`-prefixed keys of an AnomalyCLIPModule checkpoint -> AnomalyCLIP.state_dict() keys.  CLIP weights
    stored in fp16 (the reference converts with `.float()`, anomaly_clip.py:67) are up-cast."""
    out = {}
    for k, v in sd.items():
        if k.startswith("net."):
            k = k[4:]
        elif "." in k and k.split(".")[0] in ("train_loss", "roc", "auroc", "pr_curve", "average_precision", "f1", "confmat"):
            continue                                  # torchmetrics state of the LightningModule
        out[k] = v.float() if torch.is_tensor(v) and v.is_floating_point() else v
    return out


def load_into(net: torch.nn.Module, ckpt, strict: bool = True) -> Tuple[list, list]:
    """ckpt: path to a Lightning .ckpt, the loaded dict, or a bare state_dict."""
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    sd = split_lightning_state_dict(sd)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    # buffers the reference recomputes from the class names are allowed to differ in presence only
    if strict and (missing or unexpected):
        raise RuntimeError(f"checkpoint does not match the module tree: missing={missing} unexpected={unexpected}")
    return list(missing), list(unexpected)


def to_lightning_state_dict(net: torch.nn.Module) -> Dict[str, torch.Tensor]:
    return {"net." + k: v.detach().cpu() for k, v in net.state_dict().items()}
