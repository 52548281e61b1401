"""max / rms error of acx_gemm_tn_x6 and of the f32 MFMA weight-gradient kernel against fp64 over seeds (conv shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops
DEV = "cuda"
gn, gl = 32, 16
for cin, cout, tiles in ((128, 512, 64), (512, 128, 64), (256, 1024, 64), (1024, 256, 64)):
    for seed in range(4):
        rows = tiles * gn * gl
        g = torch.Generator().manual_seed(1000 * seed + cin + cout + tiles)
        dy = (torch.randn(rows, cout, generator=g) * 0.3).to(DEV)
        x = torch.randn(rows, cin, generator=g).to(DEV)
        y6 = ops.gemm_tn_x6(ops.split_bf16x3(dy), ops.split_bf16x3(x), conv=True, gn=gn, gl=gl, cin=cin)
        y32 = ops.gemm_tn(dy, x, conv=True, gn=gn, gl=gl, cin=cin)
        xp = torch.zeros(tiles, gn + 2, gl + 2, cin, dtype=torch.float64, device=DEV)
        xp[:, 1:-1, 1:-1] = x.double().view(tiles, gn, gl, cin)
        cols = torch.cat([xp[:, kh:kh + gn, kw:kw + gl] for kh in range(3) for kw in range(3)], dim=-1).reshape(rows, 9 * cin)
        ref = dy.double().t() @ cols
        e6, e32 = (y6.double() - ref).abs(), (y32.double() - ref).abs()
        print(f"cin {cin} cout {cout} seed {seed}: x6 max {e6.max():.3e} rms {e6.pow(2).mean().sqrt():.3e} | f32 max {e32.max():.3e} rms {e32.pow(2).mean().sqrt():.3e} | ratio max {e6.max() / e32.max():.2f} rms {(e6.pow(2).mean() / e32.pow(2).mean()).sqrt():.2f}")
