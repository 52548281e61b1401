"""TemporalModel (reference temporal_model.py:8-77) with the axial transformer of the un-vendored
`axial_attention` dependency re-implemented on libacx kernels.

PARITY NOTE: the arithmetic of `AxialImageTransformer` lives in an UNPINNED PyPI package whose
source is not part of the reference checkout; this module follows the restatement in
oracle/axial_attention_restated.py.  Parity for this row is "unpinned".
The parameter tree keeps upstream's names (`axial_attn.pos_emb.param_{0,1}`,
`axial_attn.layers.blocks.{i}.{f,g}.net...`) so a reference checkpoint's keys line up.

Data layout: tokens are kept channels-LAST, rows ordered (tile, n, l) with E contiguous -- every
per-token op (LayerNorm, Linear) is a row op, the two axial attentions only differ in the row
stride between sequence elements, and the 3x3 convolutions over the (N=32, L=16) grid become
implicit GEMMs whose A rows are shifted token rows (acx_gemm AMAP_CONV3X3).  The reference's
test-mode re-tiling "(b n s l) d -> (b s) n l d" is folded into the projection GEMM's row gather
and undone by the classifier kernel's scatter.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from .classification_head import ClassificationHead
from .clip_vit import LayerNorm, _Linear


class _SelfAttention(nn.Module):
    def __init__(self, dim, heads, dim_heads):
        super().__init__()
        hidden = heads * dim_heads
        self.to_q = _Linear(dim, hidden, bias=False)
        self.to_kv = _Linear(dim, 2 * hidden, bias=False)
        self.to_out = _Linear(hidden, dim)


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)


class _Holder(nn.Module):
    """`PermuteToFrom` / `Deterministic` wrappers of upstream: only their attribute names matter."""

    def __init__(self, name: str, child: nn.Module):
        super().__init__()
        setattr(self, name, child)


class _ChanLayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))


class _Conv3x3(nn.Module):
    """nn.Conv2d(cin, cout, 3, padding=1)'s parameters.  `weight` has Conv2d's logical shape [cout, cin, 3, 3] (state_dict
    compatible) but lives in channels-last MEMORY, i.e. physically [cout][kh][kw][cin] -- exactly the [Cout][tap][Cin]
    operand of the implicit-GEMM kernels, so the forward reads the master weight as it stands, the weight gradient
    [cout, 9 cin] of acx_gemm_tn is written straight into the parameter's gradient view and AdamW (elementwise on the
    physical memory of p / g / m / v) needs no re-layout: 8 permute-copies of 9.4 MB per step less at the UCF config."""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cout, cin, 3, 3).normal_(std=(9 * cin) ** -0.5)
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros(cout))

    def kernel_weight(self) -> torch.Tensor:
        """[cout, 9 cin] tap-major view of the master weight (a copy only if someone replaced the parameter's memory format)."""
        w = self.weight.detach().permute(0, 2, 3, 1)
        if not w.is_contiguous():
            w = w.contiguous()
        return w.reshape(w.shape[0], -1)


def _ff(dim):
    return nn.ModuleList([_ChanLayerNorm(dim), _Conv3x3(dim, 4 * dim), nn.Identity(), _Conv3x3(4 * dim, dim)])


class _Block(nn.Module):
    def __init__(self, f, g):
        super().__init__()
        self.f = _Holder("net", f)
        self.g = _Holder("net", g)


class _PosEmb(nn.Module):
    def __init__(self, dim, n, l):
        super().__init__()
        self.param_0 = nn.Parameter(torch.randn(1, dim, n, 1))
        self.param_1 = nn.Parameter(torch.randn(1, dim, 1, l))


class AxialImageTransformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_heads, num_segments, seg_length):
        super().__init__()
        self.dim, self.depth, self.heads = dim, depth, heads
        self.e = dim_heads if dim_heads else dim // heads
        self.pos_emb = _PosEmb(dim, num_segments, seg_length)
        blocks = []
        for _ in range(depth):
            attn = [_Holder("fn", _PreNorm(dim, _SelfAttention(dim, heads, self.e))) for _ in range(2)]
            blocks.append(_Block(attn[0], attn[1]))
            blocks.append(_Block(_ff(dim), _ff(dim)))
        self.layers = _Holder("blocks", nn.ModuleList(blocks))


class TemporalModel(nn.Module):
    def __init__(self, input_size: int, emb_size: int, output_size: int, heads: int, dim_heads: Optional[int],
                 depth: int, num_segments: int, seg_length: int):
        super().__init__()
        self.input_size, self.emb_size, self.output_size = input_size, emb_size, output_size
        self.heads, self.dim_heads, self.depth = heads, dim_heads, depth
        self.num_segments, self.seg_length = num_segments, seg_length
        self.projection = _Linear(input_size, emb_size)
        self.axial_attn = AxialImageTransformer(emb_size, depth, heads, dim_heads, num_segments, seg_length)
        self.classifier = ClassificationHead(emb_size, output_size)
        # "f32": exact-f32 MFMA (parity path).  "bf16" (inference only, BASELINE configs[4]): every GEMM / implicit-GEMM
        # convolution runs on the bf16 MFMA with f32 accumulation -- bf16 weights, bf16 LayerNorm outputs and conv
        # hidden activations; the residual streams, attention, norms and the classifier stay f32.
        # "auto": f32 results; the convolutions of the feed-forward blocks (93 % of the head's flops) and their input / weight
        # gradients run as f32-ACCURATE bf16 x 6 products on the bf16 matrix cores (three bf16 planes per operand, acx_gemm_x6.h)
        # when the geometry allows (x6_convs), everything else on the f32 MFMA kernels.
        self.precision = "f32"
        self.graph = False                 # training: replay forward / backward as HIP graphs (functional._TemporalGraphs)
        self._prep = None

    def x6_convs(self) -> bool:
        """the feed-forward convolutions as bf16 x 6 products: precision "auto", a power-of-two token grid of whole 256-row tiles
        (the convolution kernel's M % 256 == 0 for any number of tiles), and channel counts the weight-gradient kernel's tiles
        divide: E % 256 == 0 (UCF-Crime / ShanghaiTech: 256 x 256 tiles) or E == 128 (XD-Violence: 256 x 128 tiles inside one tap for
        c1's gradient [512, 9 x 128], 128 x 256 tiles for c2's [128, 9 x 512]; acx_gemm_tn_x6's tile geometries)"""
        N, Lg = self.num_segments, self.seg_length
        return (self.precision == "auto" and (self.emb_size % 256 == 0 or self.emb_size == 128) and N & (N - 1) == 0 and Lg & (Lg - 1) == 0 and
                (N * Lg) % 256 == 0)

    # ---- derived weight layouts for the kernels
    def _derived(self, train: bool):
        """Static buffers of every derived layout + the segment list that fills them (ops.prep_multi: ONE launch).  Allocated
        once per (device, parameter addresses); views of the master parameters where no re-layout is needed."""
        # ONE SLOT PER `train` FLAG: captured graphs (step_graph.TrainStepGraph, functional._TemporalGraphs) hold raw pointers
        # into the train = True buffers; an evaluation call that builds the train = False layouts must never release them
        key = (bool(train), self.precision) + tuple(p.data_ptr() for p in self.parameters())
        slots = self.__dict__.setdefault("_drv", {})
        cur = slots.get(bool(train))
        if cur is not None and cur[0] == key:
            return cur[1], cur[2]
        if torch.cuda.is_current_stream_capturing():
            raise L.AcxError("TemporalModel: derived weight buffers must be created by an eager call before a HIP graph is captured")
        P, segs = {}, []
        dev = self.projection.weight.device
        E, N, Lg = self.emb_size, self.num_segments, self.seg_length
        K = self.input_size
        Kp = (K + 31) // 32 * 32
        w = self.projection.weight.detach()
        if Kp == K:
            P["proj_w"] = w
        else:                                              # zero-padded to a multiple of 32 columns (pad written once, here)
            P["proj_w"] = torch.zeros(E, Kp, dtype=torch.float32, device=dev)
            segs.append((w, P["proj_w"], E, K, K, Kp, 0))
        P["Kp"] = Kp
        pe = self.axial_attn.pos_emb
        P["pos0"] = torch.empty(N, E, dtype=torch.float32, device=dev)            # param_0 (1, E, N, 1) -> [N, E]
        P["pos1"] = torch.empty(Lg, E, dtype=torch.float32, device=dev)           # param_1 (1, E, 1, L) -> [L, E]
        segs.append((pe.param_0.detach(), P["pos0"], E, N, N, E, 1))
        segs.append((pe.param_1.detach(), P["pos1"], E, Lg, Lg, E, 1))
        if train:
            P["proj_wT"] = torch.empty(Kp, E, dtype=torch.float32, device=dev)
            if Kp != K:
                P["proj_wT"].zero_()
            segs.append((w, P["proj_wT"], E, K, K, E, 1))
        blks = self.axial_attn.layers.blocks
        He = self.heads * self.axial_attn.e
        for d in range(self.depth):
            for fg in ("f", "g"):
                sa = getattr(blks[2 * d], fg).net.fn.fn
                tq, tkv, tout = sa.to_q.weight.detach(), sa.to_kv.weight.detach(), sa.to_out.weight.detach()
                q = P[f"qkv_w{d}{fg}"] = torch.empty(3 * He, E, dtype=torch.float32, device=dev)      # [to_q ; to_kv]
                segs.append((tq, q, He, E, E, E, 0))
                segs.append((tkv, q[He:], 2 * He, E, E, E, 0))
                ff = getattr(blks[2 * d + 1], fg).net
                # conv weights: the channels-last master IS [Cout][tap = kh*3+kw][Cin] (K = 9 Cin, tap-major)
                P[f"c1_w{d}{fg}"], P[f"c2_w{d}{fg}"] = ff[1].kernel_weight(), ff[3].kernel_weight()
                P[f"g{d}{fg}"] = ff[0].g.detach().reshape(-1)
                P[f"b{d}{fg}"] = ff[0].b.detach().reshape(-1)
                if train:
                    # operands of the dX GEMMs (dX = dY W needs W^T rows as the [N, K] operand)
                    qT = P[f"qkv_wT{d}{fg}"] = torch.empty(E, 3 * He, dtype=torch.float32, device=dev)
                    segs.append((tq, qT, He, E, E, 3 * He, 1))
                    segs.append((tkv, qT[:, He:], 2 * He, E, E, 3 * He, 1))
                    oT = P[f"out_wT{d}{fg}"] = torch.empty(He, E, dtype=torch.float32, device=dev)
                    segs.append((tout, oT, E, He, He, He, 1))
                    # dX of a 3x3 convolution = a 3x3 convolution of dY with flipped taps and swapped channels:
                    # out[ci][tap'][co] = w[co][8 - tap'][ci], nine strided [Cout, Cin] -> [Cin, Cout] transposes per weight
                    for name, conv in (("c1_dx", ff[1]), ("c2_dx", ff[3])):
                        kw = conv.kernel_weight()
                        Cout, Cin = kw.shape[0], kw.shape[1] // 9
                        if kw.data_ptr() != conv.weight.data_ptr():
                            raise L.AcxError("conv weights must stay in channels-last memory for training (see _Conv3x3)")
                        o = P[f"{name}{d}{fg}"] = torch.empty(Cin, 9 * Cout, dtype=torch.float32, device=dev)
                        for t in range(9):
                            segs.append((kw.data_ptr() + 4 * t * Cin, o.data_ptr() + 4 * (8 - t) * Cout, Cout, Cin, 9 * Cin,
                                         9 * Cout, 1))
        # bf16 x 6 convolutions: three bf16 planes of every conv operand (split after the layouts above are current)
        P["_x6"] = []
        if self.x6_convs():
            for k in [k for k in P if isinstance(k, str) and k.startswith(("c1_w", "c2_w", "c1_dx", "c2_dx"))]:
                P[k + "_3"] = torch.empty((3,) + tuple(P[k].shape), dtype=torch.bfloat16, device=dev)
                P["_x6"].append((P[k], P[k + "_3"]))
        slots[bool(train)] = (key, P, segs)
        return P, segs

    def refresh_prepared(self, train: bool = True):
        """(Re)build every derived layout from the current master weights: one acx_prep_multi launch on the current stream
        (capturable -- the step graphs replay it after each optimizer update)."""
        P, segs = self._derived(train)
        ops.prep_multi(segs, device=self.projection.weight.device)
        ops.split_bf16x3_multi(P["_x6"])                     # the convolution weights' bf16 planes: one launch
        self._prep = (self._prep_key(train), P)
        return P

    def _prep_key(self, train: bool):
        return (ops.WEIGHT_EPOCH[0], self.precision, bool(train)) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def prepared(self, train: bool = False):
        """Derived layouts, rebuilt when a parameter changed (cached by optimizer epoch / tensor versions).  `train` adds the
        backward operands (transposes, flipped-tap conv weights); a training cache also serves inference callers."""
        if self._prep is not None:
            k = self._prep[0]
            if k == self._prep_key(train) or (not train and k == self._prep_key(True)):
                return self._prep[1]
        with torch.no_grad():
            P = dict(self.refresh_prepared(train))
            if self.precision == "bf16":
                P = dict(P)
                blks = self.axial_attn.layers.blocks
                for d in range(self.depth):
                    for fg in ("f", "g"):
                        sa = getattr(blks[2 * d], fg).net.fn.fn
                        P[f"out_w{d}{fg}"] = ops.cast_bf16(sa.to_out.weight.detach())
                for k in [k for k in P if k == "proj_w" or k.startswith(("qkv_w", "c1_w", "c2_w")) and "T" not in k]:
                    P[k] = ops.cast_bf16(P[k].contiguous())
        self._prep = (self._prep_key(train), P)
        return P

    def _attn(self, x_in, resid, d, fg, tiles, axis, P):
        N, Lg, E = self.num_segments, self.seg_length, self.emb_size
        pn = getattr(self.axial_attn.layers.blocks[2 * d], fg).net.fn
        bf = self.precision == "bf16"
        prec = L.PREC_BF16 if bf else L.PREC_F32
        h = ops.layernorm(x_in, pn.norm.weight, pn.norm.bias, out_dtype=torch.bfloat16 if bf else torch.float32)
        qkv = ops.gemm(h, P[f"qkv_w{d}{fg}"], prec=prec)
        o = ops.axial_attention(qkv, tiles, N, Lg, self.heads, self.axial_attn.e, axis)
        return ops.gemm(o, P[f"out_w{d}{fg}"] if bf else pn.fn.to_out.weight, bias=pn.fn.to_out.bias, residual=resid, prec=prec)

    def _ff(self, x_in, resid, d, fg, P):
        N, Lg, E = self.num_segments, self.seg_length, self.emb_size
        ff = getattr(self.axial_attn.layers.blocks[2 * d + 1], fg).net
        bf = self.precision == "bf16"
        prec, adt = (L.PREC_BF16, torch.bfloat16) if bf else (L.PREC_F32, torch.float32)
        if f"c1_w{d}{fg}_3" in P:          # bf16 x 6: ChanLayerNorm and the first convolution write their results as planes
            h3 = ops.layernorm(x_in, P[f"g{d}{fg}"], P[f"b{d}{fg}"], mode=L.NORM_CHAN, planes_out=True)
            u3 = ops.gemm_x6(h3, P[f"c1_w{d}{fg}_3"], bias=ff[1].bias, act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=E,
                             planes_out=True)
            return ops.gemm_x6(u3, P[f"c2_w{d}{fg}_3"], bias=ff[3].bias, residual=resid, amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=4 * E)
        h = ops.layernorm(x_in, P[f"g{d}{fg}"], P[f"b{d}{fg}"], mode=L.NORM_CHAN, out_dtype=adt)
        u = ops.gemm(h, P[f"c1_w{d}{fg}"], bias=ff[1].bias, act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=E,
                     prec=prec, out_dtype=adt)
        return ops.gemm(u, P[f"c2_w{d}{fg}"], bias=ff[3].bias, residual=resid, amap=L.AMAP_CONV3X3, gn=N, gl=Lg,
                        cin=4 * E, prec=prec)

    def forward(self, features: torch.Tensor, segment_size: int, test_mode: bool,
                a_sub: Optional[torch.Tensor] = None, tile_table: Optional[torch.Tensor] = None) -> torch.Tensor:
        """features [rows, input_size] -> scores [rows, 1] (temporal_model.py:42-77).  `a_sub` fuses the
        caller's re-centring (anomaly_clip.py:143,201) into the projection GEMM's A staging.  `tile_table` (test mode,
        int32 [tiles, 2] on the device: base row, row stride between segments) replaces `segment_size` for a BATCH of videos
        with different segment sizes: tile t gathers rows base + n * stride + l and its scores are scattered back there."""
        from . import functional as Fn
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if test_mode:
                # TemporalFn implements the TRAIN tiling "(b n l) d" only; the reference evaluates under Lightning's
                # no_grad (validation_step / test_step), so a differentiable test-mode pass is not part of the path.
                raise RuntimeError("TemporalModel(test_mode=True) must run under torch.no_grad(): the differentiable "
                                   "path implements the training tiling only (temporal_model.py:55-60)")
            if self.precision not in ("f32", "auto"):
                raise RuntimeError("the bf16 head is an inference path; training runs on the f32 / f32-accurate kernels")
            return Fn.temporal_train(self, features, a_sub)
        P = self.prepared()
        N, Lg, E = self.num_segments, self.seg_length, self.emb_size
        x = features.reshape(-1, features.shape[-1])
        rows = x.shape[0]
        if x.shape[1] != P["Kp"]:
            raise ValueError("pad the temporal input to prepared()['Kp'] columns (ops.concat_features does)")
        x = x.contiguous()
        tiles = rows // (N * Lg)
        seg = segment_size if test_mode else 0
        if tile_table is not None and not test_mode:
            raise ValueError("tile_table is a test-mode (no_grad) argument")
        amap = L.AMAP_TILETABLE if tile_table is not None else (L.AMAP_TESTTILE if test_mode else L.AMAP_IDENTITY)
        x0 = ops.gemm(x, P["proj_w"], bias=self.projection.bias, a_sub=a_sub, amap=amap, gn=N, gl=Lg, seg=max(seg, 1),
                      pos0=P["pos0"], pos1=P["pos1"], prec=L.PREC_BF16 if self.precision == "bf16" else L.PREC_F32,
                      tile_table=tile_table)
        x1 = x2 = x0
        for d in range(self.depth):
            y1 = self._attn(x2, x1, d, "f", tiles, 0, P)      # long-term: along the N segments
            y2 = self._attn(y1, x2, d, "g", tiles, 1, P)      # short-term: along the L frames
            x1 = self._ff(y2, y1, d, "f", P)
            x2 = self._ff(x1, y2, d, "g", P)
        c = self.classifier
        s = ops.cls_head(x1, x2, c.layer_norm.weight, c.layer_norm.bias, c.linear.weight, c.linear.bias, N, Lg, seg,
                         tile_table=tile_table)
        return s.view(-1, 1)
