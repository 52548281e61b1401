import sys, os
sys.path.insert(0, "/root/repo")
import torch
from anomalyclip_amd import ops, _lib as L
M = 1078
for name, N, K, act, res in (("qkv", 1536, 512, 0, 0), ("out", 512, 512, 0, 1), ("fc", 2048, 512, 1, 0), ("proj", 512, 2048, 0, 1)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    kw = {}
    if res: kw["residual"] = torch.randn(M, N, device="cuda")
    if act: kw["act"] = L.ACT_QUICKGELU
    out = torch.empty(M, N, device="cuda")
    for _ in range(5): ops.gemm(a, w, bias=b, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.gemm(a, w, bias=b, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print(f"{name:5s} M={M} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
