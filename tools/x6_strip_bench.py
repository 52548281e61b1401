"""Cost of the column strips of gemm_x6_p4_kernel (ACX_OPT_X6_STRIP_TAIL) relative to a whole 256 x 256 tile, and the ViT's
frames/s by frames per launch under the three routings of the ACX_PREC_F32X6 driver.  HIP-event timed.

    python tools/x6_strip_bench.py [--vit] [--gemm]
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import ops, _lib as L
from anomalyclip_amd import init_weights as IW

ap = argparse.ArgumentParser()
ap.add_argument("--vit", action="store_true")
ap.add_argument("--gemm", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--chunks", default="32,64,128,160,256,512")
args = ap.parse_args()
dev = torch.device("cuda", 0)
di = 0
ncu = torch.cuda.get_device_properties(dev).multi_processor_count


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if args.gemm or not args.vit:
    print(f"# strips vs whole tiles, {ncu} CUs; t0 = whole tiles only, t2 / t3 = the last round as 128- / 64-column strips")
    print("N K tiles rem t0_ms t2_ms t3_ms auto_ms  c2 c1 (strip round / whole round)")
    for N, K in ((768, 768), (768, 3072), (2304, 768), (3072, 768)):
        tn = N // 256
        w3 = ops.split_bf16x3((torch.randn(N, K, device=dev) * K ** -0.5), panel=True)
        b = torch.randn(N, device=dev)
        for rem_target in (16, 32, 60, 64, 79, 100, 128, 158, 200):
            tm = (ncu + rem_target + tn - 1) // tn
            M = tm * 256
            xt = tm * tn
            rem = xt - (xt // ncu) * ncu
            a3 = ops.split_bf16x3(torch.randn(M, K, device=dev) * 0.5, panel=True)
            out = torch.empty(M, N, device=dev)
            ts = []
            for mode in (0, 2, 3, 1):
                ops.set_x6_strip_tail(di, mode)
                ts.append(timed(lambda: ops.gemm_x6(a3, w3, bias=b, out=out, split_k=False, panels=3), args.iters))
            ops.set_x6_strip_tail(di, 1)
            # one full round alone
            Mf = (ncu // tn) * 256
            tfull = timed(lambda: ops.gemm_x6(a3, w3, bias=b, out=out, split_k=False, panels=3, M=Mf), args.iters) * ncu / ((ncu // tn) * tn)
            r2 = (2 * rem + ncu - 1) // ncu
            r3 = (4 * rem + ncu - 1) // ncu
            c2 = (ts[1] - tfull) / tfull / r2
            c1 = (ts[2] - tfull) / tfull / r3
            print(f"{N} {K} {xt} {rem} {ts[0]:.4f} {ts[1]:.4f} {ts[2]:.4f} {ts[3]:.4f}  {c2:.3f} {c1:.3f}  (round {tfull:.4f} ms)")

if args.vit:
    from anomalyclip_amd.components.clip_vit import VisionTransformer
    g = IW.VIT_B16
    torch.manual_seed(0)
    for prec_name, min_tiles, strip in (("auto x6 always, strips", 1, 1), ("auto x6 always, whole tiles", 1, 0), ("auto x6 >= 512 tiles (round 5)", 512, 0),
                                        ("f32 MFMA", 1, 1)):
        vit = VisionTransformer(g.image_resolution, g.vision_patch_size, g.vision_width, g.vision_layers, g.vision_width // 64, g.embed_dim,
                                precision="f32" if prec_name.startswith("f32") else "auto").to(dev)
        ops.set_x6_min_tiles(di, min_tiles)
        ops.set_x6_strip_tail(di, strip)
        row = []
        for chunk in [int(c) for c in args.chunks.split(",")]:
            frames = torch.randn(chunk, 3, 224, 224, device=dev)
            vit.chunk = chunk
            ms = timed(lambda: vit(frames), max(3, args.iters // 4))
            row.append(f"{chunk}: {chunk / ms * 1e3:7.0f}")
        print(f"{prec_name:34s} frames/s by frames per launch  " + "  ".join(row))
    ops.set_x6_min_tiles(di, 512)
    ops.set_x6_strip_tail(di, 1)
