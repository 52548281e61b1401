"""f32-accurate GEMM from three bf16 planes per operand (acx_gemm_desc.pairs = 6) against the native f32 MFMA kernels: error vs fp64
and time at the ViT shapes (development probe)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops, _lib as L
from bench import _event_time

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
# accuracy on a shape fp64 can check
M, N, K = 4096, 768, 768
a = torch.randn(M, K, generator=g, device=dev) * 1.5 + 0.3
a[:, 5] *= 40.0                                              # a massive-activation channel
w = torch.randn(N, K, generator=g, device=dev) * 0.05
b = torch.randn(N, generator=g, device=dev)
ref = a.double() @ w.double().t() + b.double()
h = L.ctx(torch.cuda.current_device())
L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, 1), h)
y32 = ops.gemm(a, w, bias=b)
a3, w3 = ops.split_bf16x3(a), ops.split_bf16x3(w)
y6 = ops.gemm_x6(a3, w3, bias=b)
ybf = ops.gemm(ops.cast_bf16(a), ops.cast_bf16(w), bias=b, prec=L.PREC_BF16)
scale = (a.double().abs() @ w.double().abs().t())
for name, y in (("native f32", y32), ("bf16 x 6", y6), ("plain bf16", ybf)):
    e = (y.double() - ref).abs()
    print(f"{name:12s} max|err| / max|ref| = {float(e.max() / ref.abs().max()):.3e}   max err / sum|a||w| = {float((e / scale).max()):.3e}")
rec = a3.float().sum(0)
print("split reconstruction max rel err", float(((rec - a).abs() / a.abs().clamp_min(1e-30)).max()))
L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, 512), h)
# time at the ViT shapes, 512 frames
M = 512 * 197
for name, N, K, act, res in (("qkv", 2304, 768, 0, 0), ("out", 768, 768, 0, 1), ("fc", 3072, 768, 1, 0), ("proj", 768, 3072, 0, 1)):
    a = torch.randn(M, K, generator=g, device=dev)
    w = torch.randn(N, K, generator=g, device=dev) * 0.05
    bias = torch.randn(N, generator=g, device=dev)
    x = torch.randn(M, N, generator=g, device=dev) if res else None
    out = x if res else torch.empty(M, N, device=dev)
    kw = dict(bias=bias, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=x, out=out)
    t32 = _event_time(lambda: ops.gemm(a, w, **kw), 5)
    a3, w3 = ops.split_bf16x3(a), ops.split_bf16x3(w)
    t6 = _event_time(lambda: ops.gemm_x6(a3, w3, **kw), 5)
    ts = _event_time(lambda: ops.split_bf16x3(a, out=a3), 5)
    fl = 2.0 * M * N * K
    print(f"{name:5s} native f32 {t32 * 1e3:7.3f} ms {fl / t32 / 1e12:6.1f} TF | x6 {t6 * 1e3:7.3f} ms {fl / t6 / 1e12:6.1f} TF-equiv "
          f"({6 * fl / t6 / 1e12:6.0f} bf16 TF) | split of A {ts * 1e3:6.3f} ms", flush=True)
