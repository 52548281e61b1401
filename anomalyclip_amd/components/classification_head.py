"""ClassificationHead (reference classification_head.py:4-15): LayerNorm -> Linear(E,1) -> Sigmoid.
Executed fused with the reversible-stream mean by acx_cls_head (see TemporalModel)."""
import torch
from torch import nn

from .clip_vit import LayerNorm, _Linear
from .. import ops


class ClassificationHead(nn.Module):
    def __init__(self, emb_size: int, n_classes: int = 1):
        super().__init__()
        if n_classes != 1:
            raise ValueError("AnomalyCLIP uses a single-logit head (anomaly_clip.py:94)")
        self.layer_norm = LayerNorm(emb_size)
        self.linear = _Linear(emb_size, n_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.reshape(-1, x.shape[-1]).contiguous()
        s = ops.cls_head(x, x, self.layer_norm.weight, self.layer_norm.bias, self.linear.weight, self.linear.bias,
                         0, 0, 0)
        return s.view(-1, 1)
