"""Thin tensor-level wrappers over the libacx C ABI: PyTorch supplies device memory and the
current HIP stream, every arithmetic step runs in the hand-written HIP kernels.  Tensors must be
contiguous CUDA(HIP) tensors; nothing here silently falls back to a torch op."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

_BF16 = torch.bfloat16


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _h(t: torch.Tensor):
    if not t.is_cuda:
        raise L.AcxError("libacx operates on device tensors only (no CPU fallback)")
    return L.ctx(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.ACX_F32
    if t.dtype == _BF16:
        return L.ACX_BF16
    raise L.AcxError(f"unsupported dtype {t.dtype}")


def cast_bf16(src: torch.Tensor) -> torch.Tensor:
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=_BF16, device=src.device)
    h = _h(src)
    L.check(L.lib().acx_cast_bf16(h, src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), h)
    return dst


def gemm(a: torch.Tensor, w: torch.Tensor, *, out: Optional[torch.Tensor] = None, M: Optional[int] = None,
         bias=None, act=L.ACT_NONE, residual=None, a_sub=None, prec=L.PREC_F32, out_dtype=torch.float32,
         amap=L.AMAP_IDENTITY, gn=0, gl=0, cin=0, seg=0, pos0=None, pos1=None) -> torch.Tensor:
    """out[M,N] = epilogue(amap(a)[M,K] @ w[N,K]^T); see include/acx.h acx_gemm_desc."""
    assert a.dim() == 2 and w.dim() == 2 and a.is_contiguous() and w.is_contiguous()
    N, K = w.shape
    if M is None:
        M = a.shape[0]
    if amap != L.AMAP_CONV3X3:
        assert a.shape[1] == K, (a.shape, w.shape)
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    d = L.GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = a.stride(0), w.stride(0), out.stride(0)
    d.a_dtype, d.c_dtype, d.prec = _dt(a), _dt(out), prec
    if (prec == L.PREC_BF16) != (w.dtype == _BF16):
        raise L.AcxError("weight dtype does not match the requested MFMA precision")
    d.bias, d.act = _ptr(bias), act
    d.residual, d.ldr = _ptr(residual), (residual.stride(0) if residual is not None else 0)
    d.a_sub = _ptr(a_sub)
    d.amap, d.gn, d.gl, d.cin, d.seg = amap, gn, gl, cin, seg
    d.pos0, d.pos1 = _ptr(pos0), _ptr(pos1)
    h = _h(a)
    L.check(L.lib().acx_gemm(h, C.byref(d), _stream()), h)
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *, eps=1e-5, mode=L.NORM_LAYER,
              out_dtype=torch.float32, rows: Optional[int] = None, ldx: Optional[int] = None) -> torch.Tensor:
    D = w.numel()
    if rows is None:
        rows = x.numel() // D
        ldx = D
    y = torch.empty(rows, D, dtype=out_dtype, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_layernorm(h, x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), y.data_ptr(), D, _dt(y), rows, D,
                                  eps, mode, _stream()), h)
    return y


def attention(qkv: torch.Tensor, batch: int, L_: int, heads: int, causal: bool) -> torch.Tensor:
    assert qkv.is_contiguous() and qkv.shape == (batch * L_, 3 * heads * 64)
    out = torch.empty(batch * L_, heads * 64, dtype=torch.float32, device=qkv.device)
    h = _h(qkv)
    L.check(L.lib().acx_attention(h, qkv.data_ptr(), qkv.stride(0), out.data_ptr(), out.stride(0), batch, L_, heads,
                                  int(causal), _stream()), h)
    return out


def text_directions(text: torch.Tensor, ncentroid: torch.Tensor, normal_id: int) -> torch.Tensor:
    Cc, D = text.shape
    dirs = torch.empty(Cc - 1, D, dtype=torch.float32, device=text.device)
    h = _h(text)
    L.check(L.lib().acx_text_directions(h, text.data_ptr(), ncentroid.data_ptr(), dirs.data_ptr(), Cc, D, normal_id,
                                        _stream()), h)
    return dirs


def selector_project(x: torch.Tensor, ncentroid: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    x = x.reshape(-1, x.shape[-1])
    assert x.is_contiguous()
    rows, D = x.shape
    C1 = dirs.shape[0]
    raw = torch.empty(rows, C1, dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_selector_project(h, x.data_ptr(), ncentroid.data_ptr(), dirs.data_ptr(), raw.data_ptr(), rows, D,
                                         C1, _stream()), h)
    return raw


def bn_stats(raw: torch.Tensor):
    rows, C1 = raw.shape
    st = torch.empty(3, C1, dtype=torch.float32, device=raw.device)
    h = _h(raw)
    L.check(L.lib().acx_bn_stats(h, raw.data_ptr(), rows, C1, st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
                                 _stream()), h)
    return st[0], st[1], st[2]


def selector_bn(raw: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, eps=1e-5,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows, C1 = raw.shape
    if out is None:
        out = torch.empty(rows, C1, dtype=torch.float32, device=raw.device)
    h = _h(raw)
    L.check(L.lib().acx_selector_bn(h, raw.data_ptr(), mean.data_ptr(), var.data_ptr(), out.data_ptr(), out.stride(0),
                                    rows, C1, eps, _stream()), h)
    return out


def axial_attention(qkv: torch.Tensor, tiles: int, gn: int, gl: int, heads: int, e: int, axis: int) -> torch.Tensor:
    assert qkv.is_contiguous() and qkv.shape == (tiles * gn * gl, 3 * heads * e)
    out = torch.empty(tiles * gn * gl, heads * e, dtype=torch.float32, device=qkv.device)
    h = _h(qkv)
    L.check(L.lib().acx_axial_attention(h, qkv.data_ptr(), out.data_ptr(), tiles, gn, gl, heads, e, axis, _stream()), h)
    return out


def cls_head(x1, x2, ln_w, ln_b, lin_w, lin_b, gn: int, gl: int, seg: int) -> torch.Tensor:
    rows, E = x1.shape
    scores = torch.empty(rows, dtype=torch.float32, device=x1.device)
    h = _h(x1)
    L.check(L.lib().acx_cls_head(h, x1.data_ptr(), x2.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), lin_w.data_ptr(),
                                 lin_b.data_ptr(), scores.data_ptr(), rows, E, gn, gl, seg, _stream()), h)
    return scores


def class_probs(sim: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
    rows, C1 = sim.shape
    probs = torch.empty_like(sim)
    h = _h(sim)
    L.check(L.lib().acx_class_probs(h, sim.data_ptr(), scores.data_ptr(), probs.data_ptr(), rows, C1, _stream()), h)
    return probs


def colsum_(acc: torch.Tensor, x: torch.Tensor) -> None:
    x = x.reshape(-1, x.shape[-1])
    assert x.is_contiguous()
    h = _h(x)
    L.check(L.lib().acx_colsum(h, x.data_ptr(), acc.data_ptr(), x.shape[0], x.shape[1], _stream()), h)


def prompt_embed(prefix, ctxv, suffix, pos: Optional[torch.Tensor], n_ctx: int) -> torch.Tensor:
    Cc, _, W = prefix.shape
    Lc = 1 + n_ctx + suffix.shape[1]
    out = torch.empty(Cc, Lc, W, dtype=torch.float32, device=prefix.device)
    h = _h(prefix)
    L.check(L.lib().acx_prompt_embed(h, prefix.data_ptr(), ctxv.data_ptr(), suffix.data_ptr(), _ptr(pos), out.data_ptr(),
                                     Cc, n_ctx, Lc, W, int(ctxv.dim() == 2), _stream()), h)
    return out


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    assert x.dim() == 2 and x.is_contiguous() and idx.dtype == torch.int64 and idx.is_cuda
    out = torch.empty(idx.numel(), x.shape[1], dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_gather_rows(h, x.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), x.shape[1], _stream()), h)
    return out


def add_bcast(x: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    assert x.is_contiguous() and p.is_contiguous()
    out = torch.empty_like(x)
    h = _h(x)
    L.check(L.lib().acx_add_bcast(h, x.data_ptr(), p.data_ptr(), out.data_ptr(), x.numel() // p.numel(), p.numel(),
                                  _stream()), h)
    return out


def concat_features(logits: torch.Tensor, x: torch.Tensor, ncentroid: torch.Tensor, Kp: int) -> torch.Tensor:
    rows, C1 = logits.shape
    D = x.shape[-1]
    out = torch.empty(rows, Kp, dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_concat_features(h, logits.data_ptr(), x.data_ptr(), ncentroid.data_ptr(), out.data_ptr(), rows, C1,
                                        D, Kp, _stream()), h)
    return out
