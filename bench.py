#!/usr/bin/env python
"""bench.py -- headline benchmark of the AnomalyCLIP hot path on MI355X.

Metric (BASELINE.json): frames/sec encoded + anomaly-scored (whole node), ViT-B/16 224^2.
Workload (configs[2] of BASELINE.json / SURVEY.md 8d config 3): one STEP = one synthetic clip of
512 frames (1,512,3,224,224) f32 already resident in HBM -> `AnomalyCLIP.forward(test_mode=True,
load_from_features=False)`: CLIP ViT-B/16 encode of the clip (one 512-frame launch by default; --vit-chunk 256
reproduces the config's batch 256),
text encoder (recomputed every step like the reference, anomaly_clip.py:136), selector, axial
temporal transformer (one S=1 tile), classifier, then the eval post-processing
softmax(similarity)*score (anomaly_clip_module.py:474-477).  Random-init weights of the UCF-Crime
configuration (no checkpoints/network in this environment), data synthetic.

    python bench.py --gpus N --steps K --warmup W [--precision f32|bf16]
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Multi-GPU: clips shard across ranks (one process per GPU), no data-path collective in the eval
path => weak scaling; value = clips of all ranks / max-over-ranks time.

One JSON line on rank 0, with `roofline` (dominant kernel = the f32 MFMA GEMM; HIP-event timed
inside the timed region by libacx's launch timer) and `cpu_baseline` (oracle on host cores,
bounded sample, rank 0 at N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FRAMES_PER_CLIP = 512
VIT_GFLOP_PER_FRAME = 35.127          # 34.895 blocks + 0.231 patch + 0.001 proj
ATTN_GFLOP_PER_FRAME = 1.43           # QK^T + PV part of the above (runs in acx_attention, not acx_gemm)
GEMM_GFLOP_PER_FRAME = VIT_GFLOP_PER_FRAME - ATTN_GFLOP_PER_FRAME
HEAD_GFLOP_PER_TILE = 10.360          # UCF head per 512-feature tile
TEXT_GFLOP_PER_CALL = 83.43           # text encoder at 14 classes
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def build_net(precision, device, vit_chunk=256):
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts
    hc = IW.UCF_HEAD
    tab = lookup_prompts(key="ucf")
    toks = torch.tensor(tab["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key="ucf", emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads,
                      dim_heads=None, num_segments=32, seg_length=16, concat_features=False, normal_id=hc.normal_id,
                      stride=1, load_from_features=False, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7,
                      ncrops=1, num_topk=3, num_bottomk=3, n_ctx=8, shared_context=False, ctx_init="",
                      precision=precision, vit_chunk=vit_chunk)
    sd = IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, seed=0)
    net.load_state_dict(sd, strict=True)
    return net.to(device).eval(), sd, toks.argmax(-1), hc


def usable_cores():
    """host cores this process may actually use: min(affinity, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(sd, eot, hc):
    """Oracle (CPU restatement of the reference path) on the host cores, bounded sample."""
    from oracle import anomalyclip_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        # ViT leg: find a frame count worth ~12 s
        f = torch.randn(8, 3, 224, 224, generator=g)
        O.vit_forward(sd, f[:2])                       # warm-up
        t0 = time.perf_counter()
        O.vit_forward(sd, f)
        dt8 = time.perf_counter() - t0
        n = int(max(8, min(512, 12.0 / (dt8 / 8))))
        n -= n % 8
        f = torch.randn(n, 3, 224, 224, generator=g)
        t0 = time.perf_counter()
        for i in range(0, n, 32):
            O.vit_forward(sd, f[i:i + 32])
        t_vit = (time.perf_counter() - t0) / n
        # head leg: one 512-feature tile incl. text encoder, selector, temporal, post-processing
        feats = torch.randn(1, 1, 512, 512, generator=g) * 0.3
        nc = torch.zeros(512)
        t0 = time.perf_counter()
        sim, sc = O.anomaly_clip_forward_test(sd, hc, feats, nc, eot, 8, 1)
        O.eval_postprocess(sim, sc, 512)
        t_head = (time.perf_counter() - t0) / 512
    return {"value": round(1.0 / (t_vit + t_head), 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU f32 restatement): ViT-B/16 on {n} frames in batches of 32 "
                      f"({t_vit * 1e3:.1f} ms/frame) + one 512-frame head tile incl. text encoder "
                      f"({t_head * 512 * 1e3:.0f} ms/tile)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vit-chunk", type=int, default=512, help="frames per ViT launch (default: the whole 512-frame clip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from anomalyclip_amd import _lib as L
    from anomalyclip_amd import ops

    net, sd, eot, hc = build_net(args.precision, dev, args.vit_chunk)
    g = torch.Generator(device=dev).manual_seed(2 + rank)
    frames = torch.randn(1, FRAMES_PER_CLIP, 3, 224, 224, generator=g, device=dev)      # resident in HBM
    nc = torch.zeros(512, device=dev)

    def step():
        with torch.no_grad():
            sim, sc = net(frames, None, nc, 1, True)
            return ops.class_probs(sim, sc), sc

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    h = L.ctx(local_rank)
    lib = L.lib()
    sync_all()
    lib.acx_prof_enable(h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        probs, sc = step()
    sync_all()
    dt = time.perf_counter() - t0
    lib.acx_prof_enable(h, 0)
    gflops_exec = ctypes.c_double(0.0)
    L.check(lib.acx_prof_gemm_flops(h, ctypes.byref(gflops_exec)), h)
    counts = (ctypes.c_int32 * 4)()
    tot = (ctypes.c_double * 4)()
    L.check(lib.acx_prof_collect(h, counts, tot), h)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(sc).all() and torch.isfinite(probs).all()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = FRAMES_PER_CLIP * args.steps * world / dt
        # roofline of the dominant kernel (acx_gemm): algorithmic flops per launch / avg launch time.
        # per step the GEMM kernel runs the ViT GEMMs of 512 frames, the text encoder's GEMMs and the
        # head's GEMMs/convs; algorithmic flops = SURVEY 8(d) per-unit figures x units per step.
        text_attn = 14 * 12 * 8 * (2 * 2 * 77 * 77 * 64) / 1e9
        gemm_gflop_step = (GEMM_GFLOP_PER_FRAME * FRAMES_PER_CLIP + (TEXT_GFLOP_PER_CALL - text_attn)
                           + HEAD_GFLOP_PER_TILE - 0.025)
        n_gemm, ms_gemm = counts[0], tot[0]
        avg_ms = ms_gemm / max(n_gemm, 1)
        # The last ViT layer is evaluated only where its output is consumed (CLS token, clip/model.py:285), so
        # the GEMM kernel EXECUTES fewer flops than the reference's dense formulation: the roofline uses the
        # flops the launches really computed (2*M*N*K summed by libacx), the dense figure is reported beside it.
        gemm_gflop_exec_step = gflops_exec.value / 1e9 / args.steps
        flop_per_launch = gemm_gflop_exec_step * args.steps / max(n_gemm, 1)  # GFLOP per launch (average)
        achieved = flop_per_launch / avg_ms if avg_ms > 0 else 0.0           # GFLOP/ms == TFLOP/s
        peak = PEAK_TFLOPS[args.precision]
        # HBM traffic of the dominant kernel: PMC counters cannot be read in-process; they are collected by
        # tools/profile_bench.sh (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
        # FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/.
        traffic = None
        pmc_file = os.path.join(REPO, "profiles", "r01_bench_f32_pmc.json")
        if args.precision == "f32" and args.vit_chunk == 512 and os.path.exists(pmc_file):
            try:
                traffic = round(json.load(open(pmc_file))["gemm"]["hbm_bytes_per_launch"])
            except Exception:
                traffic = None
        out = {
            "metric": "frames/sec encoded + anomaly-scored (whole node), ViT-B/16 224^2",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "bf16(mfma)/f32(acc,attention)", "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic 224x224 RGB frames, ViT-B/16 encode + "
                                   "text encoder + selector + axial temporal head + eval post-processing; "
                                   "step = one 512-frame clip per GPU, UCF-Crime head config, random-init weights",
                       "frames_per_step_per_gpu": FRAMES_PER_CLIP, "vit_chunk": args.vit_chunk, "precision": args.precision},
            "roofline": {"bound": "mfma", "kernel": "acx_gemm (gemm_f32_w8_kernel / gemm_kernel, v_mfma_f32_32x32x2_f32)"
                         if args.precision == "f32" else "acx_gemm (gemm_bf16_ring_kernel / gemm_bf16_dma_kernel / gemm_kernel, v_mfma_f32_32x32x16_bf16)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": "profiles/r01_bench_f32_pmc.json (rocprofv3 PMC, bytes per launch)" if traffic else None, "launches": int(n_gemm), "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_gflop_per_launch": round(flop_per_launch, 3),
                         "gemm_gflop_per_step_executed": round(gemm_gflop_exec_step, 1),
                         "gemm_gflop_per_step_dense_reference": round(gemm_gflop_step, 1)},
            "kernel_time_ms_per_step": {"gemm": round(tot[0] / args.steps, 3), "attention": round(tot[1] / args.steps, 3),
                                        "norm_rows": round(tot[2] / args.steps, 3), "other": round(tot[3] / args.steps, 3)},
            "end_to_end_tflops": round((VIT_GFLOP_PER_FRAME * FRAMES_PER_CLIP + TEXT_GFLOP_PER_CALL + HEAD_GFLOP_PER_TILE)
                                       * world / ms_per_step, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, eot, hc)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
