"""Host-side mirror of the reference's CLIP transformer modules (clip/model.py:174-290): same
module tree and parameter names (so reference / Lightning checkpoints load with load_state_dict),
but `forward` hands raw device pointers to libacx instead of calling torch ops.

Reference mapping
    LayerNorm                 clip/model.py:174-180
    ResidualAttentionBlock    clip/model.py:188-217
    Transformer               clip/model.py:220-230
    VisionTransformer         clip/model.py:233-290
The CLIP image encoder is frozen in AnomalyCLIP (anomaly_clip_module.py:68-69), so only a forward
exists for it (no autograd through the ViT).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
from torch import nn

from .. import _lib as L
from .. import ops

# "auto" (the default) = "f32x6": f32 everywhere, the four large GEMMs of a ViT layer as f32-ACCURATE products on the bf16 matrix
# cores (three bf16 planes per operand, six cross products, f32 accumulation: acx_gemm_desc.pairs, acx_gemm_x6.h) -- lower
# error against fp64 than the f32 MFMA kernels at ~0.6 of their time (gfx950's f32 MFMA peak is 1/16 of its bf16 peak); problems
# too small for that kernel (few-frame launches, the text tower, CLS-only rows) run on the f32 MFMA kernels.  "f32": the f32 MFMA
# kernels everywhere (v_mfma_f32_32x32x2_f32: an fmaf chain).  "bf16": bf16 operands, not a parity path.
# "bf16x3" (opt-in, NOT f32-accurate): the plane kernels with the three leading cross products only (ACX_PREC_F32X3)
# "f16x3" (opt-in): TWO fp16 planes per operand, three exact products -- f32-MFMA-level results for operands inside fp16's range
PRECISIONS = {"auto": L.PREC_F32X6, "f32": L.PREC_F32, "bf16": L.PREC_BF16, "f32x6": L.PREC_F32X6, "bf16x3": L.PREC_F32X3,
              "f16x3": L.PREC_F16X3}
F16X3_WSCALE = 1024.0                                  # include/acx.h ACX_F16X3_WSCALE


class LayerNorm(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        return ops.layernorm(x.reshape(-1, shp[-1]).contiguous(), self.weight, self.bias).view(shp)


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's names (weight [out,in], bias [out])."""

    def __init__(self, fin: int, fout: int, bias: bool = True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout)) if bias else None
        nn.init.normal_(self.weight, std=fin ** -0.5)


class _MHA(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names."""

    def __init__(self, width: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = _Linear(width, width)
        nn.init.normal_(self.in_proj_weight, std=width ** -0.5)


class _MLP(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.c_fc = _Linear(width, 4 * width)
        self.c_proj = _Linear(4 * width, width)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width: int, heads: int):
        super().__init__()
        self.attn = _MHA(width)
        self.ln_1 = LayerNorm(width)
        self.mlp = _MLP(width)
        self.ln_2 = LayerNorm(width)


class Transformer(nn.Module):
    """Holds `resblocks`; executes through acx_transformer_forward (in place on a [rows, W] f32 buffer)."""

    def __init__(self, width: int, layers: int, heads: int, causal: bool = False):
        super().__init__()
        if width != heads * 64:
            raise ValueError("libacx attention kernels are built for head dim 64 (every CLIP ViT/text tower)")
        self.width, self.layers, self.heads, self.causal = width, layers, heads, causal
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])
        self._cache = None

    # -- weight table handed to the C ABI (host array of device pointers), rebuilt when parameters change
    def _key(self, prec):
        return (prec, ops.WEIGHT_EPOCH[0]) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def block_table(self, prec: int):
        key = self._key(prec)
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        arr = (L.BlockWeights * self.layers)()
        keep: List[torch.Tensor] = []
        for i, b in enumerate(self.resblocks):
            w = arr[i]
            w.ln1_w, w.ln1_b = b.ln_1.weight.data_ptr(), b.ln_1.bias.data_ptr()
            w.ln2_w, w.ln2_b = b.ln_2.weight.data_ptr(), b.ln_2.bias.data_ptr()
            w.in_proj_w, w.in_proj_b = b.attn.in_proj_weight.data_ptr(), b.attn.in_proj_bias.data_ptr()
            w.out_proj_w, w.out_proj_b = b.attn.out_proj.weight.data_ptr(), b.attn.out_proj.bias.data_ptr()
            w.fc_w, w.fc_b = b.mlp.c_fc.weight.data_ptr(), b.mlp.c_fc.bias.data_ptr()
            w.proj_w, w.proj_b = b.mlp.c_proj.weight.data_ptr(), b.mlp.c_proj.bias.data_ptr()
            if prec in (L.PREC_BF16, L.PREC_F32X6, L.PREC_F32X3, L.PREC_F16X3):
                for name, p in (("in_proj_w_bf16", b.attn.in_proj_weight), ("out_proj_w_bf16", b.attn.out_proj.weight),
                                ("fc_w_bf16", b.mlp.c_fc.weight), ("proj_w_bf16", b.mlp.c_proj.weight)):
                    # bf16 mode: one rounded copy; f32x6: the three planes hi | mid | lo of the f32 weight, each in K-panel
                    # layout [K / 32][N][32] (ACX_BF16X3P: what acx_transformer_forward's bf16 x 6 products read)
                    if prec == L.PREC_F16X3:            # two fp16 planes of 2^10 w (K-panel layout)
                        t = ops.split_f16x2(p.detach(), panel=True, scale=F16X3_WSCALE)
                    else:
                        t = ops.cast_bf16(p.detach()) if prec == L.PREC_BF16 else ops.split_bf16x3(p.detach(), panel=True)
                    keep.append(t)
                    setattr(w, name, t.data_ptr())
        self._cache = (key, (arr, keep))
        return self._cache[1]

    def forward_(self, x: torch.Tensor, batch: int, seq: int, prec: int = L.PREC_F32) -> torch.Tensor:
        """In-place forward on x [batch*seq, W] (f32, contiguous)."""
        assert x.is_contiguous() and x.dtype == torch.float32 and x.shape == (batch * seq, self.width)
        lib = L.lib()
        nbytes = lib.acx_transformer_workspace_bytes_prec(self.width, batch * seq, prec)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        arr, _keep = self.block_table(prec)
        h = ops._h(x)
        L.check(lib.acx_transformer_forward(h, x.data_ptr(), batch, seq, self.width, self.heads, self.layers,
                                            int(self.causal), prec, arr, ws.data_ptr(), nbytes, ops._stream()), h)
        return x


class _Conv(nn.Module):
    def __init__(self, width: int, patch: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch))
        nn.init.normal_(self.weight, std=(3 * patch * patch) ** -0.5)


class VisionTransformer(nn.Module):
    """CLIP ViT image encoder; `forward(frames[F,3,R,R]) -> [F, output_dim]` runs acx_vit_encode in
    chunks of `chunk` frames (workspace is allocated once per chunk size and reused)."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int,
                 output_dim: int, precision: str = "auto", chunk: int = 512, streams: int = 1):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.width, self.layers, self.heads = width, layers, heads
        self.conv1 = _Conv(width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.precision = precision
        self.chunk = chunk
        # streams = 2 (opt-in): every chunk runs as TWO half chunks on two side streams with their own workspaces -- one half's
        # memory-bound launches (LayerNorm, attention) then run beside the other half's matrix-bound GEMMs (+1-2 % frames/s at 512
        # frames, tools/probes/vit_two_streams.py).  Same kernels, same per-row arithmetic: the features are those of two launches of
        # chunk / 2 frames.  Not the default: per-launch timings of overlapping kernels (bench.py's roofline) stop meaning anything, and
        # the two extra HIP streams share the process's four hardware queues with every other stream -- created BEFORE a training step
        # graph they put its text stream onto the main stream's queue (configs[1] step 10.8 -> 15.8 ms, measured in bench.py; the step
        # graph now picks a stream that runs beside its caller's -- ops.side_stream_beside, profiles/r06_stream_queues.txt).
        self.streams = streams
        self._wcache = None
        self._ws: Optional[torch.Tensor] = None
        self._ws2: Optional[torch.Tensor] = None
        self._side = None

    @torch.no_grad()
    def f16x3_range_report(self) -> dict:
        """Worst-case magnitudes of every operand the opt-in precision "f16x3" stores as fp16 planes, PROVEN from the weights alone
        (whatever the frames are, the patch embedding excepted: its A operand is the pixels themselves):
          LayerNorm outputs   |y_i| <= |gamma_i| sqrt(D - 1) + |beta_i|          (a unit-variance row has no element above sqrt(D - 1))
          attention outputs   convex combinations of V rows: |v_j| <= ln1_bound * ||W_v,j||_1 + |b_v,j|
          QuickGELU hiddens   |x sigmoid(1.702 x)| <= |x| <= ln2_bound * ||W_fc,j||_1 + |b_fc,j|
          weight planes       2^10 |w|
        against fp16's largest finite number 65504.  `safe` says every bound is below it; `margin` is the smallest ratio 65504 / bound."""
        D = self.width
        lim = 65504.0
        worst = {"weight_planes": 0.0, "layernorm_out": 0.0, "attention_out": 0.0, "mlp_hidden": 0.0}

        def ln_bound(ln):
            return float((ln.weight.abs() * (D - 1) ** 0.5 + ln.bias.abs()).max())
        mats = [self.conv1.weight.reshape(D, -1)]
        for b in self.transformer.resblocks:
            l1, l2 = ln_bound(b.ln_1), ln_bound(b.ln_2)
            wv, bv = b.attn.in_proj_weight[2 * D:], b.attn.in_proj_bias[2 * D:]
            worst["layernorm_out"] = max(worst["layernorm_out"], l1, l2)
            worst["attention_out"] = max(worst["attention_out"], float((wv.abs().sum(1) * l1 + bv.abs()).max()))
            worst["mlp_hidden"] = max(worst["mlp_hidden"], float((b.mlp.c_fc.weight.abs().sum(1) * l2 + b.mlp.c_fc.bias.abs()).max()))
            mats += [b.attn.in_proj_weight, b.attn.out_proj.weight, b.mlp.c_fc.weight, b.mlp.c_proj.weight]
        worst["weight_planes"] = F16X3_WSCALE * max(float(m.abs().max()) for m in mats)
        margin = min(lim / max(v, 1e-30) for v in worst.values())
        return {"bounds": worst, "fp16_max": lim, "margin": margin, "safe": margin > 1.0}

    def _weights(self, prec: int):
        key = (prec, ops.WEIGHT_EPOCH[0]) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._wcache is not None and self._wcache[0] == key:
            return self._wcache[1]
        keep = []
        w = L.VitWeights()
        conv = self.conv1.weight.detach().reshape(self.width, -1).contiguous()
        proj_t = self.proj.detach().t().contiguous()
        keep += [conv, proj_t]
        w.conv1_w, w.proj_t = conv.data_ptr(), proj_t.data_ptr()
        if prec == L.PREC_BF16:
            cb, pb = ops.cast_bf16(conv), ops.cast_bf16(proj_t)
            keep += [cb, pb]
            w.conv1_w_bf16, w.proj_t_bf16 = cb.data_ptr(), pb.data_ptr()
        elif prec in (L.PREC_F32X6, L.PREC_F32X3, L.PREC_F16X3):
            # the patch embedding as a bf16 x 6 product: the weight's three planes in K-panel layout (ACX_BF16X3P)
            cb = ops.split_f16x2(conv, panel=True, scale=F16X3_WSCALE) if prec == L.PREC_F16X3 else ops.split_bf16x3(conv, panel=True)
            keep.append(cb)
            w.conv1_w_bf16 = cb.data_ptr()
        w.class_embedding = self.class_embedding.data_ptr()
        w.positional_embedding = self.positional_embedding.data_ptr()
        w.ln_pre_w, w.ln_pre_b = self.ln_pre.weight.data_ptr(), self.ln_pre.bias.data_ptr()
        w.ln_post_w, w.ln_post_b = self.ln_post.weight.data_ptr(), self.ln_post.bias.data_ptr()
        arr, keep2 = self.transformer.block_table(prec)
        w.blocks = C.cast(arr, C.POINTER(L.BlockWeights))
        keep += [arr, keep2]
        self._wcache = (key, (w, keep))
        return self._wcache[1]

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 4 and x.shape[1] == 3 and x.shape[2] == x.shape[3] == self.input_resolution
        x = x.contiguous().float()
        prec = PRECISIONS[self.precision]
        lib = L.lib()
        d = L.VitDesc(self.input_resolution, self.patch_size, self.width, self.layers, self.heads, self.output_dim, prec)
        w, _keep = self._weights(prec)
        F = x.shape[0]
        out = torch.empty(F, self.output_dim, dtype=torch.float32, device=x.device)
        chunk = min(self.chunk, F)
        nbytes = lib.acx_vit_workspace_bytes(C.byref(d), chunk)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != x.device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        h = ops._h(x)
        if self.streams == 2 and chunk >= 32 and x.is_cuda:
            return self._forward_two_streams(x, out, d, w, chunk, nbytes, h)
        s = ops._stream()
        for f0 in range(0, F, chunk):
            n = min(chunk, F - f0)
            L.check(lib.acx_vit_encode(h, C.byref(d), C.byref(w), x[f0:f0 + n].data_ptr(), n, out[f0:f0 + n].data_ptr(),
                                       self._ws.data_ptr(), self._ws.numel(), s), h)
        return out

    def _forward_two_streams(self, x, out, d, w, chunk, nbytes, h):
        lib = L.lib()
        F = x.shape[0]
        half = (chunk + 1) // 2
        if self._ws2 is None or self._ws2.numel() < nbytes or self._ws2.device != x.device:
            self._ws2 = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        if self._side is None or self._side[0].device != x.device:
            s0 = torch.cuda.Stream(device=x.device)
            self._side = (s0, ops.side_stream_beside(s0, x.device))      # two streams on two hardware queues, or they serialise
        cur = torch.cuda.current_stream()
        wss = (self._ws, self._ws2)
        for st in self._side:
            st.wait_stream(cur)                     # the frames (and the workspaces' previous readers) are ordered before
        k = 0
        for f0 in range(0, F, half):
            n = min(half, F - f0)
            st, ws = self._side[k % 2], wss[k % 2]
            with torch.cuda.stream(st):
                L.check(lib.acx_vit_encode(h, C.byref(d), C.byref(w), x[f0:f0 + n].data_ptr(), n, out[f0:f0 + n].data_ptr(),
                                           ws.data_ptr(), ws.numel(), st.cuda_stream), h)
            k += 1
        for st in self._side:
            cur.wait_stream(st)
        return out
