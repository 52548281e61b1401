"""Micro-benchmark of acx_gemm on the ViT-B/16 shapes (M = 197*frames).  HIP-event timed."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--prec", default="f32")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--shapes", default="qkv,out,fc,proj")
ap.add_argument("--epi", type=int, default=0)   # 1: the in-model epilogues (residual on out/proj, QuickGELU on fc)
ap.add_argument("--rounds", type=int, default=1)
ap.add_argument("--cbf16", action="store_true")   # bf16 C (the in-model qkv / fc outputs of the bf16 mode)
ap.add_argument("--inplace", action="store_true")   # residual aliases C (the in-model residual stream: x = x + linear(h))
ap.add_argument("--custom", default="")   # e.g. 1536x768,4608x768  (NxK)
args = ap.parse_args()
M = 197 * args.frames
SH = {"qkv": (2304, 768), "out": (768, 768), "fc": (3072, 768), "proj": (768, 3072)}
dev = "cuda"
prec = L.PREC_F32 if args.prec == "f32" else L.PREC_BF16
names = args.shapes.split(",") if not args.custom else []
for c in filter(None, args.custom.split(",")):
    n_, k_ = c.split("x"); SH[c] = (int(n_), int(k_)); names.append(c)
for name in names * args.rounds:
    N, K = SH[name]
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    if prec == L.PREC_BF16:
        w = ops.cast_bf16(w)
        a = ops.cast_bf16(a)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if args.cbf16 else torch.float32)
    kw = {}
    if args.epi and name in ("out", "proj"):
        kw["residual"] = out if args.inplace else torch.randn(M, N, device=dev)
    if args.epi and name == "fc":
        kw["act"] = L.ACT_QUICKGELU
    for _ in range(2):
        ops.gemm(a, w, bias=b, out=out, prec=prec, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.gemm(a, w, bias=b, out=out, prec=prec, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"{name:5s} M={M} N={N} K={K}  {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
