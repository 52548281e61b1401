"""CPU oracle: a plain restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this file.  The product path (anomalyclip_amd/)
never imports it and fails loudly when the HIP extension is missing.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Arithmetic is torch-CPU (fp32 by default, fp64 on request for error budgeting); the functions
take a flat state_dict with the reference's parameter names (see anomalyclip_amd/init_weights.py).

Pinning: tests/golden/*.npz were produced by running the REFERENCE's own modules (imported from
/root/reference through tests/golden/ref_harness.py) on seeded inputs; tests/test_oracle_golden.py
checks this oracle against every one of them.  Exception: row a7 (axial transformer) -- the
arithmetic lives in the un-vendored, unpinned PyPI dependency `axial_attention`; the fixtures for
it come from oracle/axial_attention_restated.py executed under the reference's TemporalModel
==> PARITY UNPINNED for a7 (see that file's header and DESIGN.md).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- CLIP blocks (a1/a2)
def layer_norm(x, w, b, eps=1e-5):
    """clip/model.py:174-180 (fp32 LayerNorm)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def quick_gelu(x):
    """clip/model.py:183-185."""
    return x * torch.sigmoid(1.702 * x)


def mha(x, in_w, in_b, out_w, out_b, n_head, causal):
    """nn.MultiheadAttention as used by clip/model.py:192,206-212 (packed in_proj, no dropout,
    need_weights=False).  x: (B, L, W) batch-first here (the reference runs LND; same math)."""
    B, L, W = x.shape
    hd = W // n_head
    qkv = x @ in_w.t() + in_b
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(B, L, n_head, hd).transpose(1, 2)
    k = k.view(B, L, n_head, hd).transpose(1, 2)
    v = v.view(B, L, n_head, hd).transpose(1, 2)
    s = (q * (hd ** -0.5)) @ k.transpose(-1, -2)
    if causal:  # clip/model.py:386-392 additive -inf above the diagonal
        m = torch.full((L, L), float("-inf"), dtype=x.dtype).triu_(1)
        s = s + m
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, L, W)
    return o @ out_w.t() + out_b


def resblock(x, sd: SD, p: str, n_head: int, causal: bool):
    """clip/model.py:214-217."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    x = x + mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], n_head, causal)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = quick_gelu(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
    return x + (h @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"])


def _num_layers(sd: SD, prefix: str) -> int:
    n = 0
    while f"{prefix}resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


def vit_forward(sd: SD, frames: torch.Tensor, prefix: str = "image_encoder.",
                return_tokens: bool = False):
    """clip/model.py:266-290.  frames (F,3,R,R) -> (F, embed_dim)."""
    w = sd[prefix + "conv1.weight"]
    width, _, ps, _ = w.shape
    n_head = width // 64
    x = F.conv2d(frames, w, stride=ps)                       # :267
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)    # :268-269  token = gy*grid+gx
    cls = sd[prefix + "class_embedding"].expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1)                           # :270-277
    x = x + sd[prefix + "positional_embedding"]              # :278
    x = layer_norm(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"])  # :279
    tp = prefix + "transformer."
    for i in range(_num_layers(sd, tp)):
        x = resblock(x, sd, f"{tp}resblocks.{i}.", n_head, causal=False)
    if return_tokens:
        return x
    x = layer_norm(x[:, 0, :], sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"])  # :285
    return x @ sd[prefix + "proj"]                           # :287-288


def prompt_assemble(sd: SD) -> torch.Tensor:
    """coop.py:74-90 (class_token_position == "end")."""
    ctx = sd["prompt_learner.ctx"]
    pre, suf = sd["prompt_learner.token_prefix"], sd["prompt_learner.token_suffix"]
    if ctx.dim() == 2:
        ctx = ctx.unsqueeze(0).expand(pre.shape[0], -1, -1)
    return torch.cat([pre, ctx, suf], dim=1)


def text_forward(sd: SD, prompts: torch.Tensor, eot_idx: torch.Tensor, n_head: int) -> torch.Tensor:
    """text_encoder.py:14-25.  prompts (C,77,W); eot_idx = tokenized_prompts.argmax(-1)."""
    x = prompts + sd["text_encoder.positional_embedding"]
    tp = "text_encoder.transformer."
    for i in range(_num_layers(sd, tp)):
        x = resblock(x, sd, f"{tp}resblocks.{i}.", n_head, causal=True)
    x = layer_norm(x, sd["text_encoder.ln_final.weight"], sd["text_encoder.ln_final.bias"])
    x = x[torch.arange(x.shape[0]), eot_idx.long()]
    return x @ sd["text_encoder.text_projection"]


def text_features(sd: SD, eot_idx: torch.Tensor, n_head: int) -> torch.Tensor:
    """anomaly_clip.py:217-221."""
    return text_forward(sd, prompt_assemble(sd), eot_idx, n_head)


# --------------------------------------------------------------------------- selector (a3/a4)
def selector_directions(text_feats, ncentroid, normal_id):
    """selector_model.py:44-59: drop the Normal row, re-centre, L2-normalise."""
    t = torch.cat((text_feats[:normal_id], text_feats[normal_id + 1:]), dim=0)
    t = t - ncentroid
    return t / t.norm(dim=-1, keepdim=True)


def selector_logits(x, text_feats, ncentroid, normal_id, running_mean, running_var,
                    training: bool, eps=1e-5, momentum=0.1):
    """selector_model.py:40-66 + BatchNorm1d(C-1, affine=False) semantics.
    Returns (logits, new_running_mean, new_running_var)."""
    x = x.reshape(-1, x.shape[-1])
    d = selector_directions(text_feats, ncentroid, normal_id)
    raw = (x - ncentroid) @ d.t()
    if training:
        mean = raw.mean(0)
        var_b = raw.var(0, unbiased=False)
        n = raw.shape[0]
        new_rm = (1 - momentum) * running_mean + momentum * mean
        new_rv = (1 - momentum) * running_var + momentum * raw.var(0, unbiased=True) if n > 1 else running_var
        return (raw - mean) / torch.sqrt(var_b + eps), new_rm, new_rv
    return (raw - running_mean) / torch.sqrt(running_var + eps), running_mean, running_var


def _topk_first_index_order(v: torch.Tensor, k: int, largest: bool) -> torch.Tensor:
    """Deterministic top-k used as the oracle's tie rule: ties broken by LOWER index first.
    torch.topk's tie order is implementation-defined (SURVEY section 7 hard parts); off ties
    this equals torch.topk exactly, which is what the golden fixtures pin."""
    key = -v if largest else v
    # stable sort ascending on key -> ties keep index order
    idx = torch.sort(key, dim=-1, stable=True)[1]
    return idx[..., :k]


def select_idx(logits, labels, mask, normal_id, num_segments, seg_length, k, largest: bool):
    """selector_model.py:119-158 (largest=True) / :227-266 (largest=False).
    logits (B, N*L, C-1); labels (B,); mask (B, N) of {0,1}.  Returns (idx_abn, idx_nor) int64."""
    b, t, c = logits.shape
    s = logits.view(b, num_segments, seg_length, c).sum(dim=2)              # :122-125
    fill = -1e6 if largest else 1e6
    drop = torch.where(mask.unsqueeze(2).expand(-1, -1, c) == 0,
                       torch.ones_like(s) * fill, s)                        # :127-130
    al = labels[: b // 2]
    al = torch.where(al > normal_id, al - 1, al)                            # :135-136
    a = drop[: b // 2]
    col = a.gather(2, al.view(-1, 1, 1).expand(-1, num_segments, 1)).squeeze(2)  # own-class column
    idx_abn = _topk_first_index_order(col, k, largest)                       # :139-150
    n = drop[b // 2:].sum(dim=2)                                            # :152-153
    idx_nor = _topk_first_index_order(n, k, largest)                         # :154-156
    return idx_abn, idx_nor


def gather_segments(logits, idx, num_segments, seg_length):
    """selector_model.py:160-225: gather the selected segments' (L, C-1) logit blocks."""
    b, t, c = logits.shape
    v = logits.view(b, num_segments, seg_length, c)
    g = v.gather(1, idx.view(b, -1, 1, 1).expand(-1, -1, seg_length, c))
    return g.reshape(-1, c)


def selector_train(x, text_feats, labels, ncentroid, normal_id, running_mean, running_var,
                   topk_mask, bottomk_mask, num_segments, seg_length, num_topk, num_bottomk):
    """selector_model.py:32-99 train branch with the mask passed in explicitly (the reference
    draws it from the CPU RNG at :101-117; the host keeps doing that)."""
    logits, rm, rv = selector_logits(x, text_feats, ncentroid, normal_id, running_mean,
                                     running_var, training=True)
    c = logits.shape[-1]
    lg = logits.view(-1, num_segments * seg_length, c)
    ia, in_ = select_idx(lg, labels, topk_mask, normal_id, num_segments, seg_length, num_topk, True)
    idx_topk = torch.cat((ia, in_), 0)
    logits_topk = gather_segments(lg, idx_topk, num_segments, seg_length)
    ba, bn = select_idx(lg, labels, bottomk_mask, normal_id, num_segments, seg_length, num_bottomk, False)
    idx_bottomk = torch.cat((ba, bn), 0)
    logits_bottomk = gather_segments(lg, idx_bottomk, num_segments, seg_length)
    return logits, logits_topk, logits_bottomk, ia, in_, ba, rm, rv


# --------------------------------------------------------------------------- temporal (a6/a7/a8)
def chan_layer_norm_last(x, g, b, eps=1e-5):
    """ChanLayerNorm (axial_attention) on channels-LAST data: eps is added to the std."""
    mu = x.mean(-1, keepdim=True)
    std = ((x - mu) ** 2).mean(-1, keepdim=True).sqrt()
    return (x - mu) / (std + eps) * g.view(-1) + b.view(-1)


def axial_self_attention(x, sd: SD, p: str, heads: int, e: int, axis: int):
    """PermuteToFrom(PreNorm(SelfAttention)) on channels-last x (T, N, L, D); axis 1 attends
    along N (long-term, perm [0,3,2,1]), axis 2 along L (short-term, perm [0,2,3,1])."""
    T, N, L, D = x.shape
    h = layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
    q = h @ sd[p + "fn.to_q.weight"].t()
    kv = h @ sd[p + "fn.to_kv.weight"].t()
    k, v = kv.chunk(2, dim=-1)
    if axis == 1:
        q, k, v = (z.transpose(1, 2) for z in (q, k, v))     # (T, L, N, He)
    S = q.shape[2]

    def split(z):
        return z.reshape(z.shape[0], z.shape[1], S, heads, e).transpose(2, 3)  # (T, A, H, S, e)

    q, k, v = split(q), split(k), split(v)
    dots = (q @ k.transpose(-1, -2)) * (e ** -0.5)
    o = torch.softmax(dots, dim=-1) @ v
    o = o.transpose(2, 3).reshape(o.shape[0], o.shape[1], S, heads * e)
    if axis == 1:
        o = o.transpose(1, 2)
    return o @ sd[p + "fn.to_out.weight"].t() + sd[p + "fn.to_out.bias"]


LEAKY_SIDE = None     # optional {conv_ff prefix p: bool tensor (T, 4D, N, L)}: see conv_ff
LEAKY_REPORT = {}


def conv_ff(x, sd: SD, p: str):
    """get_ff(): ChanLayerNorm -> Conv2d(D,4D,3,pad1) -> LeakyReLU(0.01) -> Conv2d(4D,D,3,pad1),
    evaluated on channels-last x (T,N,L,D) (conv done channels-first through F.conv2d)."""
    h = chan_layer_norm_last(x, sd[p + "0.g"], sd[p + "0.b"])
    h = h.permute(0, 3, 1, 2)
    h = F.conv2d(h, sd[p + "1.weight"], sd[p + "1.bias"], padding=1)
    side = None if LEAKY_SIDE is None else LEAKY_SIDE.get(p)
    if side is None:
        h = F.leaky_relu(h, 0.01)
    else:
        # gradient checks against a LOWER-precision path (tests/test_gpu_train.py): LeakyReLU has a kink at 0, and a
        # pre-activation within round-off of 0 lands on either side depending on the arithmetic -- a discrete change of that
        # element's derivative (1 <-> 0.01).  With LEAKY_SIDE[p] (bool, the other path's choice per element) the activation is
        # evaluated on THAT side; LEAKY_REPORT[p] records how many elements disagreed with this path's own sign and how far
        # from 0 the farthest of them was (relative to max |pre-activation|): the caller asserts "round-off of 0".
        own = h.detach() > 0
        dis = own != side
        LEAKY_REPORT[p] = (int(dis.sum()), float((h.detach().abs() * dis).max() / h.detach().abs().max()), dis.numel())
        h = torch.where(side, h, 0.01 * h)
    h = F.conv2d(h, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    return h.permute(0, 2, 3, 1)


def axial_transformer(x, sd: SD, prefix: str, depth: int, heads: int, e: int):
    """AxialImageTransformer(reversible=True) on channels-last x (T,N,L,D)."""
    p0 = sd[prefix + "pos_emb.param_0"]   # (1,D,N,1)
    p1 = sd[prefix + "pos_emb.param_1"]   # (1,D,1,L)
    x = x + p0.permute(0, 2, 3, 1)
    x = x + p1.permute(0, 2, 3, 1)
    x1, x2 = x, x
    for d in range(depth):
        a = f"{prefix}layers.blocks.{2 * d}."
        y1 = x1 + axial_self_attention(x2, sd, a + "f.net.fn.", heads, e, axis=1)
        y2 = x2 + axial_self_attention(y1, sd, a + "g.net.fn.", heads, e, axis=2)
        c = f"{prefix}layers.blocks.{2 * d + 1}."
        x1 = y1 + conv_ff(y2, sd, c + "f.net.")
        x2 = y2 + conv_ff(x1, sd, c + "g.net.")
    return (x1 + x2) / 2


def test_tile_index(rows: int, num_segments: int, seg_length: int, segment_size: int) -> torch.Tensor:
    """temporal_model.py:46-53: "(b n s l) d -> (b s) n l d".  Returns src[(b s) n l] = flat index
    into the (b n s l) ordering."""
    b = rows // (num_segments * segment_size * seg_length)
    idx = torch.arange(rows).view(b, num_segments, segment_size, seg_length)
    return idx.permute(0, 2, 1, 3).reshape(-1)


def temporal_forward(feats, sd: SD, hc, segment_size: int, test_mode: bool,
                     prefix: str = "temporal_model."):
    """temporal_model.py:42-77 + classification_head.py:11-15.  feats (rows, in) -> scores (rows,1)."""
    N, L, E = hc.num_segments, hc.seg_length, hc.emb_size
    h = feats @ sd[prefix + "projection.weight"].t() + sd[prefix + "projection.bias"]   # :43
    rows = h.shape[0]
    if test_mode:
        src = test_tile_index(rows, N, L, segment_size)
        x = h[src].view(-1, N, L, E)                                                    # :46-53
    else:
        x = h.view(-1, N, L, E)                                                         # :55-60
    x = axial_transformer(x, sd, prefix + "axial_attn.", hc.depth, hc.heads, hc.e)      # :62-66
    x = x.reshape(-1, E)
    if test_mode:                                                                       # :69-71
        out = torch.empty_like(x)
        out[src] = x
        x = out
    cp = prefix + "classifier."
    x = layer_norm(x, sd[cp + "layer_norm.weight"], sd[cp + "layer_norm.bias"])
    x = x @ sd[cp + "linear.weight"].t() + sd[cp + "linear.bias"]
    return torch.sigmoid(x)


# --------------------------------------------------------------------------- assembly (a5)
def anomaly_clip_forward_test(sd: SD, hc, image_features, ncentroid, eot_idx, text_heads,
                              segment_size=1, frames: Optional[torch.Tensor] = None,
                              text_feats: Optional[torch.Tensor] = None):
    """anomaly_clip.py:117-154.  image_features (b, ncrops, t, d) (or frames (b,t,c,h,w)).  `text_feats`: the
    (C, d) text features when the caller already evaluated them (they do not depend on the video; the reference
    recomputes them per call, :136)."""
    if frames is not None:
        b, t, c, h, w = frames.shape
        f = vit_forward(sd, frames.view(-1, c, h, w))
        d = f.shape[-1]
        # "(b ncrops n s l) d -> b ncrops (n s l) d"
        image_features = f.view(b, hc.ncrops, -1, d)
    b, ncrops, t, d = image_features.shape
    x = image_features.reshape(-1, t, d)
    tf = text_features(sd, eot_idx, text_heads) if text_feats is None else text_feats
    sim, _, _ = selector_logits(x, tf, ncentroid, hc.normal_id,
                                sd["selector_model.bn_layer.running_mean"],
                                sd["selector_model.bn_layer.running_var"], training=False)
    xc = (x - ncentroid).reshape(-1, d)
    feats = torch.cat((sim, xc), dim=-1) if hc.concat_features else xc
    scores = temporal_forward(feats, sd, hc, segment_size, True)
    sim = sim.repeat_interleave(hc.stride, dim=0)
    scores = scores.repeat_interleave(hc.stride, dim=0).view(-1)
    return sim, scores


def anomaly_clip_forward_train(sd: SD, hc, image_features, labels, ncentroid, eot_idx, text_heads,
                               topk_mask, bottomk_mask, frames: Optional[torch.Tensor] = None):
    """anomaly_clip.py:156-215 with the dropout masks passed in."""
    if frames is not None:
        b, t, c, h, w = frames.shape
        f = vit_forward(sd, frames.view(-1, c, h, w))
        image_features = f.view(b, hc.ncrops, -1, f.shape[-1])
    b, ncrops, t, d = image_features.shape
    x = image_features.reshape(-1, d)          # torch.squeeze + view(-1,d): ncrops == 1 in training
    tf = text_features(sd, eot_idx, text_heads)
    (logits, logits_topk, logits_bottomk, ia, in_, ba, rm, rv) = selector_train(
        x, tf, labels, ncentroid, hc.normal_id,
        sd["selector_model.bn_layer.running_mean"], sd["selector_model.bn_layer.running_var"],
        topk_mask, bottomk_mask, hc.num_segments, hc.seg_length, hc.num_topk, hc.num_bottomk)
    xc = x - ncentroid
    feats = torch.cat((logits, xc), dim=-1) if hc.concat_features else xc
    scores = temporal_forward(feats, sd, hc, 1, False).view(-1)
    return logits, logits_topk, scores, ia, in_, ba, rm, rv


def eval_postprocess(similarity, scores, num_labels):
    """anomaly_clip_module.py:474-483: joint class probabilities, padded frames stripped."""
    sm = torch.softmax(similarity, dim=1)
    cp = sm * scores.unsqueeze(1)
    return cp[:num_labels], scores[:num_labels]


def ncentroid_from_features(feature_list):
    """anomaly_clip_module.py:145-171: mean over every frame of every normal training video."""
    s = None
    n = 0
    for f in feature_list:
        f = f.reshape(-1, f.shape[-1])
        s = f.sum(0) if s is None else s + f.sum(0)
        n += f.shape[0]
    return s / n


# --------------------------------------------------------------------------- loss (a9)
def compute_loss(similarity, similarity_topk, labels, scores, idx_topk_abn, idx_topk_nor,
                 idx_bottomk_abn, *, normal_id, num_topk, num_segments, frames_per_segment,
                 lambda_dir_abn=1.0, lambda_dir_nor=1.0, lambda_topk_abn=1.0,
                 lambda_bottomk_abn=1.0, lambda_topk_nor=1.0, lambda_smooth=8e-4,
                 lambda_sparse=8e-3):
    """loss.py:51-195, vectorised (no per-class Python loop; same sums)."""
    B = labels.shape[0]
    C1 = similarity.shape[1]
    L, N, K = frames_per_segment, num_segments, num_topk
    al = labels[: B // 2].clone()
    al_adj = torch.where(al > normal_id, al - 1, al)                       # :82-83
    a_topk = similarity_topk[: (B // 2) * K * L].view(B // 2, K * L, C1)   # :78
    own = a_topk.gather(2, al_adj.view(-1, 1, 1).expand(-1, K * L, 1)).squeeze(2)
    ldir_abn = lambda_dir_abn * -1.0 * own.mean()                          # :87-99
    nsim = similarity[similarity.shape[0] // 2:]
    ldir_nor = lambda_dir_nor * nsim.max(dim=1)[0].mean()                  # :102-103
    sm = torch.softmax(similarity, dim=1)
    cp = sm * scores.unsqueeze(1)
    cp = torch.cat((cp[:, :normal_id], (1 - scores).unsqueeze(1), cp[:, normal_id:]), dim=1)  # :107-120
    C = C1 + 1
    cp = cp.view(-1, N, L, C)
    acp, ncp = cp[: B // 2], cp[B // 2:]

    def gat(v, idx):
        return v.gather(1, idx.view(idx.shape[0], -1, 1, 1).expand(-1, -1, L, C)).reshape(-1, C)

    a_top = torch.log(gat(acp, idx_topk_abn))
    a_bot = torch.log(gat(acp, idx_bottomk_abn))
    n_top = torch.log(gat(ncp, idx_topk_nor))
    tgt = torch.where(al_adj >= normal_id, al_adj + 1, al_adj).repeat_interleave(K * L)   # :149
    ltopk_abn = lambda_topk_abn * F.nll_loss(a_top, tgt)
    lbottomk_abn = lambda_bottomk_abn * F.nll_loss(
        a_bot, torch.full((a_bot.shape[0],), normal_id, dtype=torch.long))
    ltopk_nor = lambda_topk_nor * F.nll_loss(
        n_top, torch.full((n_top.shape[0],), normal_id, dtype=torch.long))
    abn = scores[: scores.shape[0] // 2]
    lsmooth = lambda_smooth * torch.sum((abn[1:] - abn[:-1]) ** 2)         # loss.py:10-17
    lsparse = lambda_sparse * abn.mean()                                   # loss.py:5-7
    cost = ldir_abn + ldir_nor + ltopk_abn + lbottomk_abn + ltopk_nor + lsmooth + lsparse
    return cost, ldir_abn, ldir_nor, ltopk_abn, lbottomk_abn, ltopk_nor, lsmooth, lsparse


# --------------------------------------------------------------------------- data (row f1) + a12
def start_indices(num_frames: int, num_segments: int, seg_length: int, stride: int = 1):
    """feature_dataset.py:243-278 (test mode): returns (start_indices int64, segment_size)."""
    import numpy as np
    seg_size = max(1, math.ceil(num_frames / (num_segments * seg_length * stride)))
    total = num_segments * seg_size
    return (np.arange(total, dtype=np.float64) * seg_length * stride).astype(np.int64), seg_size


def warmup_cosine_lr(base_lr: float, epoch: int, warmup_epochs: int, total_epoch: int):
    """scheduler.py:21-68 WarmupCosineAnnealingLR (warmup_powers=1, warmup_lrs=0, final_factor=0), lr after `epoch`
    scheduler steps."""
    if epoch < warmup_epochs:
        return base_lr * epoch / warmup_epochs
    prog = min((epoch - warmup_epochs) / (total_epoch - warmup_epochs), 1.0)
    return base_lr * (math.cos(math.pi * prog) + 1) / 2


# --------------------------------------------------------------------------- frame preprocessing (row f2)
def preprocess_frames_ref(frames_u8, size: int = 224):
    """src/utils/augmentations.py:21-34 with PIL directly (torchvision is not installed in this image; on PIL
    images torchvision's Resize IS `Image.resize` and CenterCrop is `Image.crop` at int(round((h-s)/2)))."""
    import numpy as np
    from PIL import Image
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    out = []
    for f in np.asarray(frames_u8):
        img = Image.fromarray(f)
        w, h = img.size
        if w <= h:
            ow, oh = size, int(size * h / w)
        else:
            oh, ow = size, int(size * w / h)
        img = img.resize((ow, oh), Image.BICUBIC)                        # gtransforms.py:89-102
        top, left = int(round((oh - size) / 2.0)), int(round((ow - size) / 2.0))
        img = img.crop((left, top, left + size, top + size))             # gtransforms.py:35-40
        t = torch.from_numpy(np.asarray(img).astype(np.float32) / np.float32(255.0)).permute(2, 0, 1)   # :373-381
        out.append((t - torch.from_numpy(mean).view(3, 1, 1)) / torch.from_numpy(std).view(3, 1, 1))    # :479-486
    return torch.stack(out)


# ---------------------------------------------------------------------------------------------------------------------
# Arithmetic of the library's `f32x6` GEMM mode (acx_gemm_desc.pairs = 6), restated on the CPU.  Not a reference function:
# the reference computes `x @ w.T` in f32 (clip/model.py:188-217 through torch); this states, in torch-CPU, what the MI355X
# kernel computes INSTEAD of an f32 FMA chain, so that tests can bound its distance to the exact product without a GPU.
def split_bf16x3(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x (f32) = hi + mid + lo, each exactly representable in bf16 (round-to-nearest-even), returned as f32 tensors.
    x - hi and (x - hi) - mid are exact in f32, so the three terms carry x's 24 significant bits."""
    x = x.float()
    hi = x.to(torch.bfloat16).float()
    r1 = x - hi
    mid = r1.to(torch.bfloat16).float()
    lo = (r1 - mid).to(torch.bfloat16).float()
    return hi, mid, lo


def matmul_bf16x6(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a [M, K] @ w [N, K]^T from the six cross products (i + j <= 2) of the bf16 planes, smallest first, as the kernel
    accumulates them.  Every bf16 x bf16 product is exact in f32; here the products are summed in f64 and rounded once, i.e.
    the result a perfect f32 accumulator would give -- the kernel's f32 accumulation over K adds its own round-off on top."""
    ah, am, al = (t.double() for t in split_bf16x3(a))
    wh, wm, wl = (t.double() for t in split_bf16x3(w))
    acc = ah @ wl.t() + am @ wm.t() + al @ wh.t() + ah @ wm.t() + am @ wh.t() + ah @ wh.t()
    return acc.float()
