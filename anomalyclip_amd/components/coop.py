"""PromptLearner (reference coop.py:10-138): learnable context vectors + frozen SOS / class-name /
EOS token embeddings, assembled as [SOS | ctx | class tokens . EOS pad] ("end" position).

The BPE tokenizer is init-time host string processing and out of scope: the
constructor takes `tokenized_prompts` (C,77) -- produced by the reference's `clip.tokenize` for
"X X X X X X X X <classname>." -- and anomalyclip_amd/data/prompts.json ships those ids for the
three label files of the reference (ucf / sht / xd)."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class PromptLearner(nn.Module):
    def __init__(self, n_cls: int, n_ctx: int, ctx_dim: int, tokenized_prompts: torch.Tensor,
                 token_embedding: torch.Tensor = None, shared_context: bool = False, ctx_init: str = ""):
        super().__init__()
        if ctx_init:
            # coop.py:19-34: context initialised from the embeddings of the given words.  The reference tokenises
            # `ctx_init` on its own; the same ids are positions 1 .. n_ctx of every tokenised prompt
            # ("<ctx_init> <classname>."), which the caller supplies (the BPE tokenizer is out of scope here).
            n_ctx = len(ctx_init.replace("_", " ").split(" "))
            if token_embedding is None:
                raise ValueError("ctx_init needs the token embedding table")
            with torch.no_grad():
                vec = token_embedding[tokenized_prompts[0, 1:1 + n_ctx].long()].float().clone()
            ctx = vec if shared_context else vec.unsqueeze(0).repeat(n_cls, 1, 1)
        else:
            ctx = torch.empty(n_ctx, ctx_dim) if shared_context else torch.empty(n_cls, n_ctx, ctx_dim)
            nn.init.normal_(ctx, std=0.02)                               # coop.py:35-43
        self.ctx = nn.Parameter(ctx)
        L = tokenized_prompts.shape[1]
        if token_embedding is not None:
            with torch.no_grad():
                emb = token_embedding[tokenized_prompts.long()]          # coop.py:57-60
        else:
            emb = torch.zeros(n_cls, L, ctx_dim)
        self.register_buffer("token_prefix", emb[:, :1, :].clone())      # coop.py:65
        self.register_buffer("token_suffix", emb[:, 1 + n_ctx:, :].clone())  # coop.py:66
        self.n_cls, self.n_ctx = n_cls, n_ctx
        self.tokenized_prompts = tokenized_prompts
        self.class_token_position = "end"

    def forward(self, positional_embedding: torch.Tensor = None) -> torch.Tensor:
        """coop.py:74-90.  With `positional_embedding` the add of text_encoder.py:15 is fused in."""
        return ops.prompt_embed(self.token_prefix, self.ctx, self.token_suffix, positional_embedding, self.n_ctx)
