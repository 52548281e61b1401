"""Development probe: the three-product mode of the plane-reuse kernel (acx_gemm_desc.pairs = 3, precision "bf16x3") -- error against
fp64 and time against the six-product form on the ViT's shapes, then the ViT encode in both modes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from anomalyclip_amd import ops, _lib as L
from bench import _event_time

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
for (M, N, K, act, res) in ((4096, 768, 768, L.ACT_NONE, False), (100864, 2304, 768, L.ACT_NONE, False), (100864, 768, 3072, L.ACT_NONE, True),
                            (100864, 3072, 768, L.ACT_QUICKGELU, False), (50432, 768, 768, L.ACT_NONE, True)):
    a = torch.randn(M, K, generator=g, device=dev)
    a[:, 5] *= 60.0
    w = torch.randn(N, K, generator=g, device=dev) * 0.03
    a3, w3 = ops.split_bf16x3(a, panel=True), ops.split_bf16x3(w, panel=True)
    r = torch.randn(M, N, generator=g, device=dev) if res else None
    outs = {}
    for pairs in (6, 3):
        fn = lambda: ops.gemm_x6(a3, w3, act=act, residual=r, panels=3, pairs=pairs)
        outs[pairs] = fn()
        t = _event_time(fn, 6)
        print(f"M {M} N {N} K {K} act {act} res {res} pairs {pairs}: {t * 1e3:.3f} ms  {2 * M * N * K / t / 1e12:.1f} TFLOP/s f32-equivalent", flush=True)
    # two fp16 planes per operand, three products (precision "f16x3"); the weights' planes hold 2^10 w
    af, wf = ops.split_f16x2(a, panel=True), ops.split_f16x2(w, panel=True, scale=1024.0)
    fnf = lambda: ops.gemm_x6(af, wf, act=act, residual=r, panels=3, pairs=3, out_scale=1.0 / 1024.0)
    outs["f16"] = fnf()
    t = _event_time(fnf, 6)
    print(f"M {M} N {N} K {K} act {act} res {res} fp16 x 2 planes, 3 products: {t * 1e3:.3f} ms  {2 * M * N * K / t / 1e12:.1f} TFLOP/s f32-equivalent", flush=True)
    if M <= 8192:
        ref = a.double() @ w.double().t()
        den = (a.double().abs() @ w.double().abs().t())
        for pairs in (6, 3, "f16"):
            e = (outs[pairs].double() - ref).abs()
            print(f"   pairs {pairs}: max err / max|ref| {float(e.max() / ref.abs().max()):.3e}   max err / sum|a||w| {float((e / den).max()):.3e}")
    else:
        e = (outs[3] - outs[6]).abs().max() / outs[6].abs().max()
        ef = (outs["f16"] - outs[6]).abs().max() / outs[6].abs().max()
        print(f"   pairs 3 vs 6: max |diff| / max|out| {float(e):.3e}   fp16 planes vs 6: {float(ef):.3e}")
    del a, w, a3, w3, r, outs

net, sd, eot, hc = B.build_net("auto", dev, 512)
vit = net.image_encoder
frames = torch.randn(512, 3, 224, 224, generator=g, device=dev)
feats = {}
for prec in ("auto", "bf16x3", "f16x3", "f32"):
    vit.precision = prec
    feats[prec] = vit(frames).clone()
    t = _event_time(lambda: vit(frames), 4)
    print(f"ViT-B/16 encode, 512 frames, precision {prec}: {512 / t:.0f} frames/s ({t * 1e3:.2f} ms)", flush=True)
ref = feats["f32"]
for prec in ("auto", "bf16x3", "f16x3"):
    d = (feats[prec] - ref).abs()
    print(f"features {prec} vs f32 MFMA path: max |diff| / max|ref| {float(d.max() / ref.abs().max()):.3e}, max elementwise rel (|ref| > 1e-2 max) "
          f"{float((d / ref.abs().clamp_min(1e-2 * float(ref.abs().max()))).max()):.3e}")

# the attention alone, six / three products, against fp64 on a few items
Bq, Lq, Hq = 512, 197, 12
qkv = torch.randn(Bq * Lq, 3 * Hq * 64, generator=g, device=dev)
q3 = ops.split_bf16x3(qkv, panel=True)
for prod in (6, 3):
    o = ops.attention_p3(q3, Bq, Lq, Hq, products=prod)
    t = _event_time(lambda: ops.attention_p3(q3, Bq, Lq, Hq, products=prod), 6)
    of = ops.unpanel(o).float().sum(0) if hasattr(ops, "unpanel") else None
    x = qkv[: 2 * Lq].double().view(2, Lq, 3, Hq, 64)
    qq, kk, vv = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) / 8.0, -1) @ vv).transpose(1, 2).reshape(2 * Lq, Hq * 64)
    err = float((of[: 2 * Lq].double() - ref).abs().max() / ref.abs().max()) if of is not None else float("nan")
    print(f"attention, {prod} products: {t * 1e3:.3f} ms per 512-frame layer, max err / max|ref| vs fp64 {err:.3e}")
