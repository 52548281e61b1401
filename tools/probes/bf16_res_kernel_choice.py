import sys; sys.path.insert(0, "/root/repo")
import torch
from anomalyclip_amd import ops, _lib as L
M = 197 * 512
h = L.ctx(0)
for ring_min in (512, 1 << 30):
    L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, ring_min), h)
    for rnd in range(2):
        for name, N, K in (("out", 768, 768), ("proj", 768, 3072)):
            a = ops.cast_bf16(torch.randn(M, K, device="cuda")); w = ops.cast_bf16(torch.randn(N, K, device="cuda") * 0.05)
            b = torch.randn(N, device="cuda"); out = torch.randn(M, N, device="cuda")
            f = lambda: ops.gemm(a, w, bias=b, out=out, residual=out, prec=L.PREC_BF16)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            if rnd: print("ring_min", ring_min, name, f"{ms:.4f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
