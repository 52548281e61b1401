"""CLIP frame preprocessing on the GPU.

Reference: `get_augmentations` (src/utils/augmentations.py:21-34) = GroupScale(224, BICUBIC) -> GroupCenterCrop(224)
-> GroupToTensor -> GroupNormalize(CLIP mean/std); on PIL images torchvision's Resize is `PIL.Image.resize`, i.e.
Pillow's two-pass 8-bit resampler (horizontal then vertical, 22-bit fixed-point coefficients, 8-bit intermediate).
The coefficient tables are computed here on the host in double precision exactly as Pillow's `precompute_coeffs` /
`normalize_coeffs_8bpc` do (published algorithm of libImaging/Resample.c); the device kernels apply them.
JPEG decoding stays on the host (out of scope)."""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache

import numpy as np
import torch

from . import _lib as L
from . import ops

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # augmentations.py:23
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)      # augmentations.py:24
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs(inSize, 0, inSize, outSize, BICUBIC) + normalize_coeffs_8bpc."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_geometry(h: int, w: int, size: int):
    """torchvision Resize(size) on (h, w): shorter side -> size (int truncation), then CenterCrop(size) offsets."""
    if w <= h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    top = int(round((oh - size) / 2.0))
    left = int(round((ow - size) / 2.0))
    return oh, ow, top, left


@lru_cache(maxsize=16)
def _tables(h: int, w: int, size: int, device_str: str):
    oh, ow, top, left = resize_geometry(h, w, size)
    hb, hk, hks = _coeffs(w, ow)
    vb, vk, vks = _coeffs(h, oh)
    dev = torch.device(device_str)
    sl_h, sl_v = slice(left, left + size), slice(top, top + size)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t(hb[sl_h]), t(hk[sl_h]), hks, t(vb[sl_v]), t(vk[sl_v]), vks


def preprocess_frames(frames_u8: torch.Tensor, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """frames_u8: [F, H, W, 3] uint8 on the device -> [F, 3, size, size] float32 (the ViT's input)."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3 and frames_u8.is_contiguous()
    F, H, W, _ = frames_u8.shape
    hb, hk, hks, vb, vk, vks = _tables(H, W, size, str(frames_u8.device))
    out = torch.empty(F, 3, size, size, dtype=torch.float32, device=frames_u8.device)
    tmp = torch.empty(F, H, size, 3, dtype=torch.uint8, device=frames_u8.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    h = ops._h(frames_u8)
    L.check(L.lib().acx_preprocess_frames(h, frames_u8.data_ptr(), out.data_ptr(), tmp.data_ptr(), hb.data_ptr(), hk.data_ptr(),
                                          hks, vb.data_ptr(), vk.data_ptr(), vks, F, H, W, size, size, m, s, ops._stream()), h)
    return out
