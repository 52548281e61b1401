"""Micro-benchmark of the implicit-GEMM 3x3 convolutions of the axial feed-forwards (generic loader path of
acx_gemm) at the UCF training shape (64 videos x 512 tokens, E=256).  HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops, _lib as L

dev = "cuda"
M = int(os.environ.get("CB_ROWS", 64 * 512))
for rnd in range(2):
    for name, cin, cout in (("conv1", 256, 1024), ("conv2", 1024, 256)):
        x = torch.randn(M, cin, device=dev)
        w = torch.randn(cout, 9 * cin, device=dev) * 0.02
        b = torch.randn(cout, device=dev)
        out = torch.empty(M, cout, device=dev)
        kw = dict(bias=b, out=out, amap=L.AMAP_CONV3X3, gn=32, gl=16, cin=cin)
        if os.environ.get("CB_RES") and name == "conv2":            # the forward conv2 carries the block's residual
            kw["residual"] = torch.randn(M, cout, device=dev)
        if os.environ.get("CB_ACT") and name == "conv1":
            kw["act"] = L.ACT_LEAKYRELU
        for _ in range(2):
            ops.gemm(x, w, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.gemm(x, w, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if rnd:
            print(f"{name} M={M} N={cout} K={9 * cin}  {ms:.4f} ms  {2.0 * M * cout * 9 * cin / ms / 1e9:.1f} TFLOP/s")
    # the same shapes through the identity (FAST) loader for reference
    for name, n, k in (("plain1", 1024, 2304), ("plain2", 256, 9216)):
        x = torch.randn(M, k, device=dev)
        w = torch.randn(n, k, device=dev) * 0.02
        out = torch.empty(M, n, device=dev)
        for _ in range(2):
            ops.gemm(x, w, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.gemm(x, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if rnd:
            print(f"{name} M={M} N={n} K={k}  {ms:.4f} ms  {2.0 * M * n * k / ms / 1e9:.1f} TFLOP/s")
