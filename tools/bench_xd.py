#!/usr/bin/env python
"""BASELINE.json configs[4]: XD-Violence-shaped long segments, bf16.
  * head: features (1, 5 crops, 512*S, 512) f32 -> text encoder (bf16) + selector + axial temporal head with bf16 MFMA GEMMs /
    implicit-GEMM convolutions (XD config: C = 7, E = 128) -> scores; features/s, GEMM TFLOP/s of the step
  * frames: 5-crop x 32-frame windows, --chunk frames (default 640 = 4 windows) per ViT-B/16 launch in bf16 mode; frames/s
One JSON line.   python tools/bench_xd.py [--S 16] [--steps 10] [--precision bf16|f32]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--windows", type=int, default=8, help="160-frame windows per frames step")
    ap.add_argument("--chunk", type=int, default=640, help="frames per ViT launch (160 = one window)")
    ap.add_argument("--head-only", action="store_true", help="skip the ViT leg (rocprof of the head alone)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts
    hc = IW.XD_HEAD
    toks = torch.tensor(lookup_prompts(key="xd")["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key="xd", emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads, dim_heads=None,
                      num_segments=32, seg_length=16, concat_features=False, normal_id=hc.normal_id, stride=1,
                      load_from_features=True, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=hc.ncrops,
                      num_topk=3, num_bottomk=3, precision=args.precision, vit_chunk=args.chunk)
    net.load_state_dict(IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, seed=0), strict=True)
    net = net.to(dev).eval()
    timer, prof = B.Timer(None, dev), B.Prof(0)
    g = torch.Generator(device=dev).manual_seed(3)
    feats = torch.randn(1, hc.ncrops, 512 * args.S, 512, generator=g, device=dev) * 0.3
    nc = torch.zeros(512, device=dev)

    def head():
        with torch.no_grad():
            net(feats, None, nc, args.S, True)
    dt = timer.run(head, args.steps, 2)
    timer.run(head, 3, 0, prof.start, prof.stop)
    gf, counts, tot = prof.collect()
    rows = hc.ncrops * 512 * args.S
    out = {"config": "configs[4] XD-Violence long segments", "precision": args.precision, "S": args.S,
           "head": {"rows_per_step": rows, "ms_per_step": round(dt / args.steps * 1e3, 3),
                    "features_per_s": round(rows * args.steps / dt, 1),
                    "gemm_tflops": round(gf / 1e9 / tot[0], 1) if tot[0] else None,
                    "gemm_frac_of_peak": round(gf / 1e9 / tot[0] / B.PEAK_TFLOPS[args.precision], 4) if tot[0] else None,
                    "kernel_ms_per_step": {"gemm": round(tot[0] / 3, 3), "attention": round(tot[1] / 3, 3),
                                           "norm_rows": round(tot[2] / 3, 3), "other": round(tot[3] / 3, 3)}}}
    if args.head_only:
        print(json.dumps(out))
        return
    frames = torch.randn(160 * args.windows, 3, 224, 224, generator=g, device=dev)

    def enc():
        with torch.no_grad():
            net.image_encoder(frames)
    dt = timer.run(enc, max(2, args.steps // 3), 1, prof.start_gemm_only, prof.stop)
    n = max(2, args.steps // 3)
    gf, counts, tot = prof.collect()
    out["frames"] = {"frames_per_launch": args.chunk, "frames_per_s": round(160 * args.windows * n / dt, 1),
                     "gemm_tflops": round(gf / 1e9 / tot[0], 1) if tot[0] else None,
                     "gemm_frac_of_peak": round(gf / 1e9 / tot[0] / B.PEAK_TFLOPS[args.precision], 4) if tot[0] else None,
                     "kernel_ms_per_window": {"gemm": round(tot[0] / n / args.windows, 3), "attention": round(tot[1] / n / args.windows, 3),
                                              "norm_rows": round(tot[2] / n / args.windows, 3)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
