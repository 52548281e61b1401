"""AnomalyCLIP model assembly -- drop-in for the reference's
`src.models.components.anomaly_clip.AnomalyCLIP` (anomaly_clip.py:17-233): same constructor keys
(configs/model/anomaly_clip_*.yaml `net:` block), same attributes the LightningModule reaches into
(`image_encoder`, `text_encoder.text_projection`, `token_embedding`, `selector_model`,
`temporal_model`, `prompt_learner`, `embedding_dim`), same `forward` signature and return tuples.

Differences forced by the environment, not by design:
  * no `clip.load` (network download + TorchScript patching, SURVEY.md section 2 "out of scope"):
    the CLIP geometry is given by `arch` and weights arrive through `load_state_dict`
    (a Lightning `.ckpt`'s `state_dict` with the `net.` prefix stripped, or
    anomalyclip_amd.init_weights for synthetic runs);
  * no BPE tokenizer: `tokenized_prompts` is looked up in anomalyclip_amd/data/prompts.json by
    class names (the three label files of the reference) or passed explicitly.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch
from torch import nn

from .. import ops
from ..init_weights import ClipGeometry, VIT_B16, TINY
from .clip_vit import VisionTransformer
from .coop import PromptLearner
from .selector_model import SelectorModel
from .temporal_model import TemporalModel
from .text_encoder import TextEncoder

_ARCH = {"ViT-B/16": VIT_B16, "tiny": TINY}
_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "prompts.json")


def _read_classnames(labels_file: str):
    import csv
    with open(labels_file) as f:
        rows = list(csv.reader(f))[1:]
    return sorted(r[1] for r in rows if len(r) >= 2)          # anomaly_clip.py:69-70


def lookup_prompts(classnames=None, key: Optional[str] = None):
    with open(_DATA) as f:
        table = json.load(f)
    if key is not None:
        return table[key]
    for v in table.values():
        if v["classnames"] == list(classnames):
            return v
    raise ValueError("no pre-tokenised prompts for these class names: pass tokenized_prompts= (from the reference's "
                     "clip.tokenize of 'X X X X X X X X <name>.')")


class _TokenEmbedding(nn.Module):
    def __init__(self, vocab: int, width: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(vocab, width).normal_(std=0.02))


class AnomalyCLIP(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        g = kwargs.get
        self.arch = g("arch", "ViT-B/16")
        self.labels_file = g("labels_file")
        self.emb_size, self.depth, self.heads, self.dim_heads = g("emb_size"), g("depth"), g("heads"), g("dim_heads")
        self.num_segments, self.seg_length = g("num_segments"), g("seg_length")
        self.concat_features = bool(g("concat_features"))
        self.normal_id, self.stride = g("normal_id"), g("stride", 1)
        self.load_from_features = g("load_from_features", True)
        self.select_idx_dropout_topk = g("select_idx_dropout_topk")
        self.select_idx_dropout_bottomk = g("select_idx_dropout_bottomk")
        self.ncrops, self.num_topk, self.num_bottomk = g("ncrops", 1), g("num_topk"), g("num_bottomk")
        # "auto" (default): f32 results; every product large enough for the plane-reuse kernel runs as an f32-accurate bf16 x 6
        # product on the bf16 matrix cores (clip_vit.PRECISIONS), everything else on the f32 MFMA kernels.  "f32": the f32 MFMA
        # kernels everywhere.  "bf16": bf16 MFMA (BASELINE configs[4]; not a parity path).  "f32x6" = "auto" (older name).
        self.precision = g("precision", "auto")
        if self.precision == "f32x6":
            self.precision = "auto"
        vit_precision = self.precision
        if self.precision in ("bf16x3", "f16x3"):
            # opt-in, NOT f32-accurate: the ViT's plane products with the three leading cross products only (ACX_PREC_F32X3: sixteen
            # significant bits per operand); the head keeps the default's arithmetic
            self.precision = "auto"
        head_precision = "f32" if self.precision == "auto" else self.precision      # text tower (too small for the bf16 x 6 kernel)
        geom = g("clip_geometry") or _ARCH[self.arch]
        if isinstance(geom, dict):
            geom = ClipGeometry(**geom)
        self.geometry = geom

        classnames = g("classnames")
        tokenized = g("tokenized_prompts")
        if classnames is None:
            if self.labels_file and os.path.isfile(self.labels_file):
                classnames = _read_classnames(self.labels_file)
            else:
                # the reference's label files are data/{ucf,sht,xd}_labels.csv (configs/data/*.yaml `labels_file`);
                # when the configured path does not exist on this machine their class names ship in prompts.json
                key = g("labels_key")
                if key is None and self.labels_file:
                    stem = os.path.basename(str(self.labels_file)).split("_")[0].lower()
                    key = stem if stem in ("ucf", "sht", "xd") else None
                classnames = lookup_prompts(key=key or "ucf")["classnames"]
        ctx_init = g("ctx_init") or ""
        if tokenized is None:
            if ctx_init:
                raise ValueError("ctx_init is set: pass tokenized_prompts= (the reference's clip.tokenize of "
                                 "'<ctx_init> <classname>.' per class); the shipped prompts.json holds the 'X X ... X' prompts only")
            tokenized = torch.tensor(lookup_prompts(classnames)["tokenized_prompts"], dtype=torch.int32)
        self.classnames = classnames

        self.embedding_dim = geom.transformer_width                      # anomaly_clip.py:72
        self.token_embedding = _TokenEmbedding(geom.vocab_size, geom.transformer_width)
        n_ctx = g("n_ctx", 8)
        self.prompt_learner = PromptLearner(len(classnames), n_ctx, geom.transformer_width, tokenized,
                                            self.token_embedding.weight.detach(), bool(g("shared_context", False)), ctx_init)
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.register_buffer("eot_index", tokenized.argmax(dim=-1).to(torch.int64), persistent=False)
        # The text transformer is CAUSAL (clip/model.py:343-349) and only each class's EOT row is read (text_encoder.py:23): the
        # positions behind the last EOT cannot influence any output or gradient, so the tower runs on the first `text_len`
        # positions only ("X X X X X X X X <name>." ends at position 11-12 of CLIP's 77: 16 rows per class instead of 77).
        # Results are those of the full-length evaluation; `text_truncate: false` restores it.
        full = int(tokenized.shape[-1])
        self.text_len = min(full, (int(tokenized.argmax(dim=-1).max()) + 1 + 3) // 4 * 4) if bool(g("text_truncate", True)) else full
        self.text_encoder = TextEncoder(geom.context_length, geom.transformer_width, geom.transformer_heads,
                                        geom.transformer_layers, geom.embed_dim, head_precision)
        self.image_encoder = VisionTransformer(geom.image_resolution, geom.vision_patch_size, geom.vision_width,
                                               geom.vision_layers, geom.vision_heads, geom.embed_dim,
                                               precision=vit_precision, chunk=g("vit_chunk", 512))
        self.selector_model = SelectorModel(classnames, self.normal_id, nn.Parameter(torch.tensor(2.6592601)),
                                            self.num_segments, self.seg_length, self.select_idx_dropout_topk,
                                            self.select_idx_dropout_bottomk, self.num_topk, self.num_bottomk)
        additional = len(classnames) - 1
        input_size = self.embedding_dim + additional * int(self.concat_features)     # anomaly_clip.py:92-93
        self.temporal_model = TemporalModel(input_size, self.emb_size, 1, self.heads, self.dim_heads, self.depth,
                                            self.num_segments, self.seg_length)
        self.temporal_model.precision = self.precision        # "auto": the feed-forward convolutions as bf16 x 6 products
        # evaluation: the prompt parameters are frozen under no_grad, so the per-video text tower of the reference
        # (anomaly_clip.py:136: recomputed for every test video) returns the same tensor every time -- it is computed once and
        # kept, keyed by the optimizer epoch and the version / address of every input of the text path (any edit of those
        # recomputes it).  `cache_text_features: false` restores the per-call evaluation.
        self.cache_text_features = bool(g("cache_text_features", True))
        self.text_eval_graph = bool(g("text_eval_graph", True))            # uncached evaluation: the tower replayed from a HIP graph
        # under data parallelism the (replicated) text encoder is evaluated class-parallel: every rank runs its block
        # of classes and the (C, E) features / their gradients are exchanged (functional.TextFeaturesFn)
        self.text_class_parallel = bool(g("text_class_parallel", True))
        # training: replay the text tower as two HIP graphs on a side stream (functional._TextGraphs; same kernels,
        # bit-identical results, ~230 fewer library calls per step and the tower runs beside the temporal model)
        self.text_graph = bool(g("text_graph", False))
        # ... and the temporal model's forward / backward as two graphs on the main stream (functional._TemporalGraphs)
        self.temporal_model.graph = bool(g("temporal_graph", False))
        self._text_cache = None

    # ------------------------------------------------------------------------------------------
    def get_text_features(self) -> torch.Tensor:
        """anomaly_clip.py:217-221 (prompt assembly + positional add fused into one kernel)."""
        from . import functional as Fn
        if torch.is_grad_enabled() and (self.prompt_learner.ctx.requires_grad or self.text_encoder.text_projection.requires_grad):
            return Fn.text_features_train(self)
        key = None
        if self.cache_text_features:
            # every input of the text path: optimizer steps through raw pointers bump WEIGHT_EPOCH, load_state_dict /
            # in-place edits bump _version, .to(device) changes data_ptr -- any of them invalidates the cache.  (The frozen
            # layers' parameter OBJECTS are listed once: Module.parameters() walks the module tree, 0.4 ms per call.)
            plist = self.text_encoder.__dict__.get("_acx_plist")
            if plist is None:
                plist = self.text_encoder.__dict__["_acx_plist"] = list(self.text_encoder.transformer.parameters())
            key = (ops.WEIGHT_EPOCH[0], self.text_len) + tuple(
                (t.data_ptr(), t._version) for t in (self.prompt_learner.ctx, self.prompt_learner.token_prefix,
                                                     self.prompt_learner.token_suffix,
                                                     self.text_encoder.text_projection,
                                                     self.text_encoder.positional_embedding,
                                                     self.text_encoder.ln_final.weight, self.text_encoder.ln_final.bias)
            ) + tuple((p.data_ptr(), p._version) for p in plist)
            if self._text_cache is not None and self._text_cache[0] == key:
                return self._text_cache[1]
        if not self.cache_text_features and self.text_eval_graph:
            tf = self._text_features_replayed()
            if tf is not None:
                return tf
        tf = self._text_features_eager()
        if self.cache_text_features:
            self._text_cache = (key, tf)
        return tf

    def _text_features_eager(self) -> torch.Tensor:
        pl = self.prompt_learner
        x = ops.prompt_embed(pl.token_prefix, pl.ctx.detach(), pl.token_suffix, self.text_encoder.positional_embedding.detach(),
                             pl.n_ctx, Lout=self.text_len)
        return self.text_encoder.encode(x, self.eot_index)

    def _text_features_replayed(self):
        """Evaluation with `cache_text_features = False` (the reference's per-call text tower, anomaly_clip.py:136): the ~170
        launches of the tower are a fixed sequence over parameters that stay where they are, so they are captured once as a
        HIP graph and replayed -- one launch from the host per call instead of ~1.4 ms of launch calls.  The layers keep derived
        weight layouts keyed by (optimizer epoch, address, version) of their parameters, so the graph is keyed the same way:
        any edit of a text-path parameter re-captures it.  Returns None where a graph cannot be used (no GPU stream capture inside another
        capture; a failed capture disables the path for this module)."""
        dev = self.prompt_learner.ctx.device
        if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return None
        plist = self.text_encoder.__dict__.get("_acx_plist")
        if plist is None:
            plist = self.text_encoder.__dict__["_acx_plist"] = list(self.text_encoder.transformer.parameters())
        te = self.text_encoder
        key = (self.text_len, te.precision, ops.WEIGHT_EPOCH[0]) + tuple(
            (t.data_ptr(), t._version) for t in (self.prompt_learner.ctx, self.prompt_learner.token_prefix,
                                                 self.prompt_learner.token_suffix, te.text_projection, te.positional_embedding,
                                                 te.ln_final.weight, te.ln_final.bias)
        ) + tuple((p.data_ptr(), p._version) for p in plist)
        g = self.__dict__.get("_text_graph")
        if g is None or g[0] != key:
            try:
                with torch.no_grad():
                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(torch.cuda.current_stream())
                    ops.prime_capture_stream(side, dev)
                    with torch.cuda.stream(side):
                        self._text_features_eager()                      # lazy workspaces, function attributes
                    torch.cuda.synchronize(dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                        out = self._text_features_eager()
                    torch.cuda.synchronize(dev)
            except Exception as e:                                       # noqa: BLE001 -- capture support is best effort
                self.text_eval_graph = False
                self.__dict__["_text_graph_error"] = repr(e)
                return None
            g = self.__dict__["_text_graph"] = (key, graph, out)
        g[1].replay()
        return g[2].clone()                                              # the static buffer is rewritten by the next replay

    def get_temporal_model_input(self, image_features, similarity, ncentroid):
        """anomaly_clip.py:223-233; returns (features, a_sub): the re-centring is either folded into
        the concat kernel or deferred to the projection GEMM's A staging."""
        x = image_features.reshape(-1, image_features.shape[-1]).contiguous()
        if self.concat_features:
            Kp = self.temporal_model.prepared()["Kp"]
            if torch.is_grad_enabled() and similarity.requires_grad:
                from .functional import ConcatFeaturesFn
                return ConcatFeaturesFn.apply(similarity, x, ncentroid, Kp), None
            return ops.concat_features(similarity.contiguous(), x, ncentroid, Kp), None
        return x, ncentroid

    def forward(self, image_features, labels, ncentroid, segment_size=1, test_mode=False):
        dev = image_features.device
        ncentroid = ncentroid.to(dev, torch.float32).contiguous()
        segment_size = int(segment_size)
        if test_mode:
            if not self.load_from_features:
                b, t, c, h, w = image_features.size()
                f = self.image_encoder(image_features.view(-1, c, h, w))                # anomaly_clip.py:119-123
                # "(b ncrops n s l) d -> b ncrops (n s l) d" is a pure view
                image_features = f.view(b, self.ncrops, -1, f.shape[-1])
            b, ncrops, t, d = image_features.shape
            x = image_features.reshape(-1, t, d).contiguous().float()
            text_features = self.get_text_features()
            similarity = self.selector_model(x, text_features, labels, ncentroid, True)
            feats, a_sub = self.get_temporal_model_input(x, similarity, ncentroid)
            scores = self.temporal_model(feats, segment_size, True, a_sub=a_sub)
            if self.stride != 1:                                                         # anomaly_clip.py:149-150
                similarity = similarity.repeat_interleave(self.stride, dim=0)
                scores = scores.repeat_interleave(self.stride, dim=0)
            return similarity, scores.view(-1)
        from . import functional as Fn
        return Fn.anomaly_clip_train_forward(self, image_features, labels, ncentroid)

    @torch.no_grad()
    def forward_test_many(self, features: torch.Tensor, rows_per_crop, segment_sizes, ncentroid):
        """Test-mode forward of SEVERAL videos in one launch sequence (the reference scores one video per step,
        configs/data/*.yaml:7 `batch_size_test: 1`, anomaly_clip_module.py:458-483; tiles of different videos are independent
        in evaluation -- BatchNorm uses its running statistics -- so they batch).

        features: [sum_v ncrops * 512 * S_v, D] rows, video after video, each video crop-major in frame order (what the
        dataset's `(1, ncrops, 512 S, D)` tensor holds); rows_per_crop[v] = 512 * S_v; segment_sizes[v] = S_v.
        Returns (similarity [rows, C-1], scores [rows]) in the same row order (stride expansion is left to the caller)."""
        if not self.load_from_features:
            raise ValueError("forward_test_many takes pre-extracted features")
        dev = features.device
        ncentroid = ncentroid.to(dev, torch.float32).contiguous()
        N, Lg = self.num_segments, self.seg_length
        x = features.reshape(-1, features.shape[-1]).contiguous().float()
        # tile table (host ints -> one small H2D copy): video v, crop c, tile s gathers rows row0 + s L + n (S_v L) + l.
        # Kept per group geometry (a test epoch repeats its groups; pinning a fresh host buffer costs ~0.1 ms per call).
        tkey = (dev, tuple(int(s_) for s_ in segment_sizes))
        tables = self.__dict__.setdefault("_tile_tables", {})
        table = tables.get(tkey)
        r0 = 0
        for rows, S in zip(rows_per_crop, segment_sizes):
            if int(rows) != N * Lg * int(S):
                raise ValueError("rows_per_crop must equal num_segments * seg_length * segment_size")
            r0 += self.ncrops * int(rows)
        if r0 != x.shape[0]:
            raise ValueError(f"features hold {x.shape[0]} rows, the video list describes {r0}")
        if table is None:
            base, stride = [], []
            r0 = 0
            for rows, S in zip(rows_per_crop, segment_sizes):
                rows, S = int(rows), int(S)
                for c in range(self.ncrops):
                    for s_ in range(S):
                        base.append(r0 + s_ * Lg)
                        stride.append(S * Lg)
                    r0 += rows
            if len(tables) >= 64:
                tables.pop(next(iter(tables)))
            table = tables[tkey] = torch.tensor(list(zip(base, stride)), dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
        text_features = self.get_text_features()
        similarity = self.selector_model(x, text_features, None, ncentroid, True)
        feats, a_sub = self.get_temporal_model_input(x, similarity, ncentroid)
        scores = self.temporal_model(feats, 1, True, a_sub=a_sub, tile_table=table)
        return similarity, scores.view(-1)
