"""Seeded random initialisation of an AnomalyCLIP state_dict with the REFERENCE's key names.

There is no network in the build/bench environment, hence no CLIP checkpoint and no published
AnomalyCLIP `.ckpt`.  Benchmarks, parity tests and golden fixtures therefore use random weights
drawn with CLIP's own initialisation scales (reference clip/model.py:352-384 and :254-264) so
activations have realistic magnitudes.  Biases / LayerNorm affines are perturbed away from their
0/1 defaults so that a kernel that drops a bias or a gain cannot pass parity.

Key layout == `AnomalyCLIP.state_dict()` of the reference (checkpoint row):
  image_encoder.*, text_encoder.*, token_embedding.weight, prompt_learner.{ctx,token_prefix,
  token_suffix}, selector_model.{logit_scale,bn_layer.*}, temporal_model.{projection,axial_attn,
  classifier}.*
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


@dataclass
class ClipGeometry:
    """Constructor arguments of the reference CLIP (clip/model.py:294-308)."""
    embed_dim: int = 512
    image_resolution: int = 224
    vision_layers: int = 12
    vision_width: int = 768
    vision_patch_size: int = 16
    context_length: int = 77
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    def as_kwargs(self) -> dict:
        return dict(embed_dim=self.embed_dim, image_resolution=self.image_resolution,
                    vision_layers=self.vision_layers, vision_width=self.vision_width,
                    vision_patch_size=self.vision_patch_size, context_length=self.context_length,
                    vocab_size=self.vocab_size, transformer_width=self.transformer_width,
                    transformer_heads=self.transformer_heads,
                    transformer_layers=self.transformer_layers)


VIT_B16 = ClipGeometry()
# tiny geometry used by fixtures (head dim stays 64 like every CLIP ViT)
TINY = ClipGeometry(embed_dim=128, image_resolution=32, vision_layers=2, vision_width=128,
                    vision_patch_size=16, context_length=77, vocab_size=49408,
                    transformer_width=128, transformer_heads=2, transformer_layers=2)


@dataclass
class HeadConfig:
    """`net.*` keys of configs/model/anomaly_clip_*.yaml that shape the head."""
    num_classes: int = 14
    normal_id: int = 7
    n_ctx: int = 8
    shared_context: bool = False
    num_segments: int = 32
    seg_length: int = 16
    emb_size: int = 256
    depth: int = 1
    heads: int = 8
    dim_heads: Optional[int] = None
    concat_features: bool = False
    num_topk: int = 3
    num_bottomk: int = 3
    select_idx_dropout_topk: float = 0.7
    select_idx_dropout_bottomk: float = 0.7
    stride: int = 1
    ncrops: int = 1

    @property
    def e(self) -> int:
        return self.dim_heads if self.dim_heads else self.emb_size // self.heads


UCF_HEAD = HeadConfig()
SHT_HEAD = HeadConfig(num_classes=18, normal_id=8, depth=2, concat_features=True)
XD_HEAD = HeadConfig(num_classes=7, normal_id=4, emb_size=128, ncrops=5)


def _n(gen, *shape, std=1.0, mean=0.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float32) * std + mean


def _resblocks(sd: Dict[str, torch.Tensor], prefix: str, width: int, layers: int, gen):
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = _n(gen, 3 * width, width, std=attn_std)
        sd[p + "attn.in_proj_bias"] = _n(gen, 3 * width, std=0.02)
        sd[p + "attn.out_proj.weight"] = _n(gen, width, width, std=proj_std)
        sd[p + "attn.out_proj.bias"] = _n(gen, width, std=0.02)
        sd[p + "ln_1.weight"] = _n(gen, width, std=0.05, mean=1.0)
        sd[p + "ln_1.bias"] = _n(gen, width, std=0.02)
        sd[p + "mlp.c_fc.weight"] = _n(gen, 4 * width, width, std=fc_std)
        sd[p + "mlp.c_fc.bias"] = _n(gen, 4 * width, std=0.02)
        sd[p + "mlp.c_proj.weight"] = _n(gen, width, 4 * width, std=proj_std)
        sd[p + "mlp.c_proj.bias"] = _n(gen, width, std=0.02)
        sd[p + "ln_2.weight"] = _n(gen, width, std=0.05, mean=1.0)
        sd[p + "ln_2.bias"] = _n(gen, width, std=0.02)


def init_vit_state_dict(geom: ClipGeometry, seed: int, prefix: str = "image_encoder.") -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    w = geom.vision_width
    scale = w ** -0.5
    ps = geom.vision_patch_size
    fan_in = 3 * ps * ps
    sd[prefix + "conv1.weight"] = _n(gen, w, 3, ps, ps, std=fan_in ** -0.5)
    sd[prefix + "class_embedding"] = _n(gen, w, std=scale)
    sd[prefix + "positional_embedding"] = _n(gen, geom.grid ** 2 + 1, w, std=scale)
    sd[prefix + "ln_pre.weight"] = _n(gen, w, std=0.05, mean=1.0)
    sd[prefix + "ln_pre.bias"] = _n(gen, w, std=0.02)
    _resblocks(sd, prefix + "transformer", w, geom.vision_layers, gen)
    sd[prefix + "ln_post.weight"] = _n(gen, w, std=0.05, mean=1.0)
    sd[prefix + "ln_post.bias"] = _n(gen, w, std=0.02)
    sd[prefix + "proj"] = _n(gen, w, geom.embed_dim, std=scale)
    return sd


def init_text_state_dict(geom: ClipGeometry, seed: int) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    w = geom.transformer_width
    sd["token_embedding.weight"] = _n(gen, geom.vocab_size, w, std=0.02)
    sd["text_encoder.positional_embedding"] = _n(gen, geom.context_length, w, std=0.01)
    _resblocks(sd, "text_encoder.transformer", w, geom.transformer_layers, gen)
    sd["text_encoder.ln_final.weight"] = _n(gen, w, std=0.05, mean=1.0)
    sd["text_encoder.ln_final.bias"] = _n(gen, w, std=0.02)
    sd["text_encoder.text_projection"] = _n(gen, w, geom.embed_dim, std=w ** -0.5)
    return sd


def init_temporal_state_dict(in_size: int, hc: HeadConfig, seed: int,
                             prefix: str = "temporal_model.") -> Dict[str, torch.Tensor]:
    """Shapes follow temporal_model.py:31-40 and the restated axial_attention (oracle/)."""
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    E, H, e = hc.emb_size, hc.heads, hc.e
    he = H * e
    sd[prefix + "projection.weight"] = _n(gen, E, in_size, std=in_size ** -0.5)
    sd[prefix + "projection.bias"] = _n(gen, E, std=0.02)
    ax = prefix + "axial_attn."
    # upstream uses randn (std 1) for the axial positional embedding; keep that
    sd[ax + "pos_emb.param_0"] = _n(gen, 1, E, hc.num_segments, 1)
    sd[ax + "pos_emb.param_1"] = _n(gen, 1, E, 1, hc.seg_length)
    for d in range(hc.depth):
        for fg in ("f", "g"):
            a = f"{ax}layers.blocks.{2 * d}.{fg}.net.fn."
            sd[a + "norm.weight"] = _n(gen, E, std=0.05, mean=1.0)
            sd[a + "norm.bias"] = _n(gen, E, std=0.02)
            sd[a + "fn.to_q.weight"] = _n(gen, he, E, std=E ** -0.5)
            sd[a + "fn.to_kv.weight"] = _n(gen, 2 * he, E, std=E ** -0.5)
            sd[a + "fn.to_out.weight"] = _n(gen, E, he, std=he ** -0.5)
            sd[a + "fn.to_out.bias"] = _n(gen, E, std=0.02)
            c = f"{ax}layers.blocks.{2 * d + 1}.{fg}.net."
            sd[c + "0.g"] = _n(gen, 1, E, 1, 1, std=0.05, mean=1.0)
            sd[c + "0.b"] = _n(gen, 1, E, 1, 1, std=0.02)
            sd[c + "1.weight"] = _n(gen, 4 * E, E, 3, 3, std=(9 * E) ** -0.5)
            sd[c + "1.bias"] = _n(gen, 4 * E, std=0.02)
            sd[c + "3.weight"] = _n(gen, E, 4 * E, 3, 3, std=(36 * E) ** -0.5)
            sd[c + "3.bias"] = _n(gen, E, std=0.02)
    sd[prefix + "classifier.layer_norm.weight"] = _n(gen, E, std=0.05, mean=1.0)
    sd[prefix + "classifier.layer_norm.bias"] = _n(gen, E, std=0.02)
    sd[prefix + "classifier.linear.weight"] = _n(gen, 1, E, std=E ** -0.5)
    sd[prefix + "classifier.linear.bias"] = _n(gen, 1, std=0.02)
    return sd


def init_anomalyclip_state_dict(geom: ClipGeometry, hc: HeadConfig, tokenized_prompts: torch.Tensor,
                                seed: int, with_image_encoder: bool = True) -> Dict[str, torch.Tensor]:
    """Full `AnomalyCLIP.state_dict()`-shaped dict.  `tokenized_prompts`: (C, 77) int token ids of
    "X X X X X X X X <classname>." (reference coop.py:53-56)."""
    sd: Dict[str, torch.Tensor] = {}
    if with_image_encoder:
        sd.update(init_vit_state_dict(geom, seed + 1))
    sd.update(init_text_state_dict(geom, seed + 2))
    gen = torch.Generator().manual_seed(seed + 3)
    C, w = hc.num_classes, geom.transformer_width
    if hc.shared_context:
        sd["prompt_learner.ctx"] = _n(gen, hc.n_ctx, w, std=0.02)
    else:
        sd["prompt_learner.ctx"] = _n(gen, C, hc.n_ctx, w, std=0.02)
    emb = sd["token_embedding.weight"][tokenized_prompts.long()]  # (C, 77, w) coop.py:57-60
    sd["prompt_learner.token_prefix"] = emb[:, :1, :].clone()           # coop.py:65
    sd["prompt_learner.token_suffix"] = emb[:, 1 + hc.n_ctx:, :].clone()  # coop.py:66
    sd["selector_model.logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    sd["selector_model.bn_layer.running_mean"] = _n(gen, C - 1, std=0.05)
    sd["selector_model.bn_layer.running_var"] = _n(gen, C - 1, std=0.02).abs() + 0.05
    sd["selector_model.bn_layer.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    in_size = geom.embed_dim + (C - 1) * int(hc.concat_features)
    sd.update(init_temporal_state_dict(in_size, hc, seed + 4))
    return sd
