"""Few-row GEMM microbenchmark: the text tower's GEMM shapes (forward and the transposed dX chain) at the row counts a
data-parallel rank sees (77 rows per class), through the few-row kernel (gemm_f32_sk_kernel) and through the tile
kernels (ACX_OPT_SK_MAX_M = 0), HIP-event timed, back-to-back launches (includes the ~1.5 us dependent-launch boundary).
Also the text attention forward / backward and the LayerNorm backward at the same row counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops, _lib as L


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


dev = torch.cuda.current_device()
for M in (77, 154, 308, 539, 1078):
    for name, N, K, act, res in (("qkv", 1536, 512, 0, 0), ("out", 512, 512, 0, 1), ("fc", 2048, 512, 0, 0), ("proj", 512, 2048, 0, 1),
                                 ("d_pre", 2048, 512, 0, 0), ("d_h2", 512, 2048, 0, 0), ("d_h1", 512, 1536, 0, 0)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        kw = {}
        if res:
            kw["residual"] = torch.randn(M, N, device="cuda")
        if act:
            kw["act"] = L.ACT_QUICKGELU
        out = torch.empty(M, N, device="cuda")
        r = {}
        for tag, lim in (("few-row", 1 << 20), ("tiles", 0)):
            ops.set_few_row_limit(dev, lim)
            r[tag] = timeit(lambda: ops.gemm(a, w, bias=b, out=out, **kw))
        ops.set_few_row_limit(dev, 320)
        print(f"{name:6s} M={M:5d} N={N:5d} K={K:5d}: few-row {r['few-row']:7.1f} us ({2.0 * M * N * K / r['few-row'] / 1e6:6.1f} TFLOP/s)   "
              f"tiles {r['tiles']:7.1f} us ({2.0 * M * N * K / r['tiles'] / 1e6:6.1f} TFLOP/s)")
    C = M // 77
    qkv = torch.randn(M, 1536, device="cuda")
    do = torch.randn(M, 512, device="cuda")
    x = torch.randn(M, 512, device="cuda")
    lw = torch.ones(512, device="cuda")
    print(f"attention fwd  C={C:2d}: {timeit(lambda: ops.attention(qkv, C, 77, 8, True)):7.1f} us")
    print(f"attention bwd  C={C:2d}: {timeit(lambda: ops.seq_attention_bwd(qkv, do, C, 1, 77, 8, 64, 1, causal=True)):7.1f} us")
    print(f"layernorm      C={C:2d}: {timeit(lambda: ops.layernorm(x, lw, lw)):7.1f} us")
    print(f"layernorm bwd+add    : {timeit(lambda: ops.layernorm_bwd(x, lw, do, need_params=False, add=x)):7.1f} us")
