"""Timing of the pairs = 6 GEMM (acx_gemm_x6.h) at the ViT-B/16 shapes of a 512-frame clip and at the head's convolutions
(development probe; ACX_LIB_PATH selects an A/B build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops, _lib as L
from bench import _event_time

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
M = 512 * 197
which = sys.argv[1] if len(sys.argv) > 1 else "vit"
if which in ("vit", "all"):
    for name, N, K, act, res in (("qkv", 2304, 768, 0, 0), ("out", 768, 768, 0, 1), ("fc", 3072, 768, 1, 0), ("proj", 768, 3072, 0, 1),
                                 ("long", 768, 12288, 0, 0)):
        a = torch.randn(M, K, generator=g, device=dev)
        w = torch.randn(N, K, generator=g, device=dev) * 0.05
        bias = torch.randn(N, generator=g, device=dev)
        x = torch.randn(M, N, generator=g, device=dev) if res else None
        out = x if res else torch.empty(M, N, device=dev)
        kw = dict(bias=bias, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=x, out=out)
        a3, w3 = ops.split_bf16x3(a), ops.split_bf16x3(w)
        t6 = _event_time(lambda: ops.gemm_x6(a3, w3, **kw), 8)
        fl = 2.0 * M * N * K
        print(f"{name:5s} x6 {t6 * 1e3:7.3f} ms {fl / t6 / 1e12:6.1f} TF-equiv ({6 * fl / t6 / 1e12:6.0f} bf16 TF)", flush=True)
        if name == "fc":
            op = torch.empty(3, M, N, dtype=torch.bfloat16, device=dev)
            t6 = _event_time(lambda: ops.gemm_x6(a3, w3, bias=bias, act=L.ACT_QUICKGELU, out=op, planes_out=True), 8)
            print(f"fc->planes x6 {t6 * 1e3:7.3f} ms {fl / t6 / 1e12:6.1f} TF-equiv ({6 * fl / t6 / 1e12:6.0f} bf16 TF)", flush=True)
if which in ("conv", "all"):
    for tiles in (8, 64):
        rows = tiles * 512
        for name, cin, cout in (("c1", 256, 1024), ("c2", 1024, 256)):
            x = torch.randn(rows, cin, generator=g, device=dev)
            w = torch.randn(cout, 9 * cin, generator=g, device=dev) * 0.02
            b = torch.randn(cout, generator=g, device=dev)
            x3, w3 = ops.split_bf16x3(x), ops.split_bf16x3(w)
            t32 = _event_time(lambda: ops.gemm(x, w, bias=b, amap=L.AMAP_CONV3X3, gn=32, gl=16, cin=cin), 8)
            t6 = _event_time(lambda: ops.gemm_x6(x3, w3, bias=b, amap=L.AMAP_CONV3X3, gn=32, gl=16, cin=cin), 8)
            fl = 2.0 * rows * cout * 9 * cin
            xi = torch.randn(3, rows, 9 * cin, generator=g, device=dev).to(torch.bfloat16)
            ti = _event_time(lambda: ops.gemm_x6(xi, w3, bias=b), 8)
            print(f"{name} rows {rows:6d}: identity-map product of the same shape: x6 {ti * 1e6:7.1f} us {fl / ti / 1e12:6.1f} TF-equiv", flush=True)
            del xi
            print(f"{name} rows {rows:6d}: f32 {t32 * 1e6:7.1f} us {fl / t32 / 1e12:6.1f} TF | x6 {t6 * 1e6:7.1f} us {fl / t6 / 1e12:6.1f} TF-equiv", flush=True)
            dy = torch.randn(rows, cout, generator=g, device=dev)
            dy3 = ops.split_bf16x3(dy)
            t32 = _event_time(lambda: ops.gemm_tn(dy, x, conv=True, gn=32, gl=16, cin=cin), 8)
            t6 = _event_time(lambda: ops.gemm_tn_x6(dy3, x3, conv=True, gn=32, gl=16, cin=cin), 8)
            ts = _event_time(lambda: ops.split_bf16x3(dy, out=dy3), 8)
            print(f"{name} dW rows {rows:6d}: f32 {t32 * 1e6:7.1f} us {fl / t32 / 1e12:6.1f} TF | x6 {t6 * 1e6:7.1f} us {fl / t6 / 1e12:6.1f} TF-equiv | split of dY {ts * 1e6:6.1f} us", flush=True)
