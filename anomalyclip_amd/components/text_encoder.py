"""TextEncoder (reference text_encoder.py:5-25): CLIP text transformer over the learned prompts,
EOT-token gather, text projection.  Same attribute names as the reference (`transformer`,
`positional_embedding`, `ln_final`, `text_projection`)."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from .clip_vit import LayerNorm, Transformer, PRECISIONS


class TextEncoder(nn.Module):
    def __init__(self, context_length: int, width: int, heads: int, layers: int, embed_dim: int,
                 precision: str = "f32"):
        super().__init__()
        self.transformer = Transformer(width, layers, heads, causal=True)   # clip/model.py:333-338,386-392
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width).normal_(std=0.01))
        self.ln_final = LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim).normal_(std=width ** -0.5))
        self.precision = precision
        self.dtype = torch.float32

    def encode(self, x: torch.Tensor, eot_index: torch.Tensor) -> torch.Tensor:
        """x: (C, L, W) prompts with the positional embedding already added (fresh buffer, modified in
        place); eot_index: (C,) int64 on device."""
        Cc, Lc, W = x.shape
        prec = PRECISIONS[self.precision]
        x2 = x.view(Cc * Lc, W)
        self.transformer.forward_(x2, Cc, Lc, prec)                         # text_encoder.py:16-18
        cache = self.__dict__.setdefault("_eot_rows", {})                    # EOT row table: built once per geometry
        key = (Cc, Lc, eot_index.data_ptr(), eot_index._version)   # in-place edits (load_state_dict) bump _version
        rows = cache.get(key)
        if rows is None:
            rows = cache[key] = (torch.arange(Cc, device=x.device, dtype=torch.int64) * Lc + eot_index).contiguous()
        eot = ops.gather_rows(x2, rows)                                      # text_encoder.py:23 (gather)
        eot = ops.layernorm(eot, self.ln_final.weight, self.ln_final.bias)   # :19 (row-wise, so gather first)
        # x @ text_projection == gemm with W = text_projection^T [E, W]; exact f32 always (tiny, trainable)
        return ops.gemm(eot, ops.transpose(self.text_projection.detach()), prec=L.PREC_F32)

    def forward(self, prompts: torch.Tensor, tokenized_prompts: torch.Tensor) -> torch.Tensor:
        x = ops.add_bcast(prompts.contiguous(), self.positional_embedding)   # text_encoder.py:15
        eot = tokenized_prompts.argmax(dim=-1).to(device=x.device, dtype=torch.int64)
        return self.encode(x, eot)
