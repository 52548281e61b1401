"""Micro-benchmark of acx_gemm_tn (weight-gradient GEMM) on the head's training shapes.  HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops

dev = "cuda"
M = 64 * 512
CASES = [("conv1 dW", 1024, 256, True), ("conv2 dW", 256, 1024, True), ("plain1 dW", 1024, 2304, False),
         ("plain2 dW", 256, 9216, False), ("qkv dW", 768, 256, False), ("proj dW", 256, 512, False)]
for rnd in range(2):
    for name, n1, cin, conv in CASES:
        a = torch.randn(M, n1, device=dev)
        b = torch.randn(M, cin, device=dev)
        kw = dict(conv=True, gn=32, gl=16, cin=cin) if conv else {}
        n2 = 9 * cin if conv else cin
        for _ in range(2):
            ops.gemm_tn(a, b, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.gemm_tn(a, b, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if rnd:
            print(f"{name:9s} M={M} N1={n1} N2={n2}  {ms:.4f} ms  {2.0 * M * n1 * n2 / ms / 1e9:.1f} TFLOP/s")
