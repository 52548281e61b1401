"""SelectorModel (reference selector_model.py:5-333): normal-direction projection, BatchNorm over
the C-1 directions, and (training) MIL top-k / bottom-k segment selection.

forward() keeps the reference signature and return values.  The Bernoulli segment-dropout mask is
drawn on the HOST from torch's CPU generator exactly like the reference (selector_model.py:101-117)
and handed to the device; everything else runs in libacx kernels."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class _BatchNormStats(nn.Module):
    """Buffer holder with nn.BatchNorm1d(affine=False)'s names."""

    def __init__(self, n: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = eps, momentum


class SelectorModel(nn.Module):
    def __init__(self, classnames: list, normal_id: int, logit_scale, num_segments: int, seg_length: int,
                 select_idx_dropout_topk: float, select_idx_dropout_bottomk: float, num_topk: int, num_bottomk: int):
        super().__init__()
        self.classnames = classnames
        self.normal_id = normal_id
        # registered, trainable and never used, as in the reference (selector_model.py:22)
        self.logit_scale = logit_scale if isinstance(logit_scale, nn.Parameter) else nn.Parameter(torch.as_tensor(logit_scale, dtype=torch.float32))
        self.num_segments, self.seg_length = num_segments, seg_length
        self.select_idx_dropout_topk = select_idx_dropout_topk
        self.select_idx_dropout_bottomk = select_idx_dropout_bottomk
        self.num_topk, self.num_bottomk = num_topk, num_bottomk
        self.bn_layer = _BatchNormStats(len(classnames) - 1)

    def generate_mask(self, batch: int):
        """selector_model.py:101-117 -- CPU RNG, one Bernoulli(1-p) per (video, segment)."""
        select_idx = torch.ones((batch, self.num_segments))
        topk = torch.bernoulli(select_idx * (1 - self.select_idx_dropout_topk))
        bottomk = torch.bernoulli(select_idx * (1 - self.select_idx_dropout_bottomk))
        if self.select_idx_dropout_topk == self.select_idx_dropout_bottomk:
            topk = bottomk
        return topk, bottomk

    def forward(self, image_features, text_features, labels, ncentroid, test_mode, masks=None):
        from . import functional as Fn
        x = image_features.reshape(-1, image_features.shape[-1])
        if test_mode:
            dirs = ops.text_directions(text_features.detach().contiguous(), ncentroid, self.normal_id)
            raw = ops.selector_project(x.contiguous(), ncentroid, dirs)
            return ops.selector_bn(raw, self.bn_layer.running_mean, self.bn_layer.running_var, self.bn_layer.eps)
        if masks is None:
            masks = self.generate_mask(x.shape[0] // (self.num_segments * self.seg_length))
        return Fn.selector_train(self, x, text_features, labels, ncentroid, masks)
