#!/usr/bin/env python
"""bench.py -- headline benchmark of the AnomalyCLIP hot path on MI355X.

Metric (BASELINE.json): frames/sec encoded + anomaly-scored (whole node), ViT-B/16 224^2.
Workload of `value` (configs[2] of BASELINE.json / SURVEY.md 8d config 3): one STEP = one synthetic clip of
512 frames (1,512,3,224,224) f32 already resident in HBM -> `AnomalyCLIP.forward(test_mode=True,
load_from_features=False)`: CLIP ViT-B/16 encode of the clip (one 512-frame launch by default; --vit-chunk 256
reproduces the config's literal batch 256), text features (frozen prompts under no_grad: cached after the first step by default;
the leg `text_recomputed_every_step` runs the text encoder in every step like the reference, anomaly_clip.py:136), selector, axial temporal transformer (one S=1 tile), classifier, then the eval post-processing
softmax(similarity)*score (anomaly_clip_module.py:474-477).  Random-init weights of the UCF-Crime configuration
(no checkpoints/network in this environment), data synthetic.

    python bench.py --gpus N --steps K --warmup W [--precision f32|bf16]
    (N>1: typed like that it re-launches itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
     --master-addr 127.0.0.1`; started by that launcher -- the driver's form -- it just runs as rank RANK of WORLD_SIZE)

`value`: clips shard across ranks (one process per GPU); the eval path has no data-path collective (the reference's
test_step is rank-local too) => weak scaling; value = frames of all ranks / max-over-ranks time.

Extra legs, each timed like the main one (barrier + synchronize on both sides, max over ranks), reported as extra
keys of the same JSON line:
  * `encode_only`, `vit_chunk_256` -- ViT encode alone; the headline step with two 256-frame ViT launches (config[2]'s
    literal batch).
  * `head` (configs[1]) -- UCF-shaped 512-d feature sequences, B = 64 videos on one GPU: forward-only and the full
    training step (forward, 7-term loss, backward, AdamW) in features/s, with the GEMM roofline of the training step.
  * `dp_train` (configs[3]) -- the same training step DATA-PARALLEL over the N ranks through RCCL: gradients in
    parallel.GradBuckets (bucketed async all-reduce, 41.7 MB), SyncBatchNorm statistics, class-parallel text
    encoder.  `strong` = 64 videos GLOBAL (64/N per rank), `weak` = 64 videos PER RANK; step time, features/s, rank 0's
    kernel-time breakdown of the step and the stand-alone all-reduce time of the gradient buffer.  Strong-scaling
    efficiency = T1 / (N * TN) from the N = 1, 2, 4, 8 lines of the driver's scaling run.

  * `config4_xd_bf16` (configs[4]) -- XD-Violence-shaped long segments through the bf16 head and 160-frame windows through
    the bf16 ViT (not the parity path; reported with the bf16 GEMM roofline fraction).
`roofline.traffic` is measured live at N = 1: two child `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE) of
this script's headline step; the committed profiles/ file is the fallback when rocprofv3 is unavailable.

One JSON line on rank 0, with `roofline` (dominant kernel = the f32 MFMA GEMM; HIP-event timed inside the timed
region by libacx's launch timer: event pairs around every acx_gemm launch of the K timed steps.  The other launch kinds
-- attention, norms, the rest: `kernel_time_ms_per_step` -- are bracketed in the two untimed initialisation steps instead:
an event pair costs ~7 us of queue time, and ~150 of them per step in the timed region took 1.1 ms off a 131.5 ms step)
and `cpu_baseline` (oracle on host cores, bounded sample, rank 0 at N=1 only).

--leg-limit S (default 1200): the headline is measured FIRST; if the secondary legs + CPU baseline have not finished S seconds later
(a normal run needs ~60 s), rank 0 prints the line from the headline alone (value, ms_per_step, the GEMM roofline of the timed steps;
`"watchdog"` says so, traffic and cpu_baseline null) and ends the process; at N > 1 the limit is at most 420 s and the other ranks leave
5 s after rank 0.  An exception that escapes a secondary leg (a collective failing because a peer is gone) ends the run the same way: the
headline-only line from rank 0, exit status 0 on every rank.  Before the headline is measured every failure is loud.
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

DTYPE_TEXT = {"auto": "f32 (bf16x6 operands, exact 24-bit split, f32 accumulate)", "f32": "f32", "bf16": "bf16(mfma)/f32(acc,attention)"}
_STATE = {}                            # the headline-only line's printer once the headline is measured (main(), __main__)
FRAMES_PER_CLIP = 512
VIT_GFLOP_PER_FRAME = 35.127          # 34.895 blocks + 0.231 patch + 0.001 proj
ATTN_GFLOP_PER_FRAME = 1.43           # QK^T + PV part of the above (runs in acx_attention, not acx_gemm)
GEMM_GFLOP_PER_FRAME = VIT_GFLOP_PER_FRAME - ATTN_GFLOP_PER_FRAME
HEAD_GFLOP_PER_TILE = 10.360          # UCF head per 512-feature tile
TEXT_GFLOP_PER_CALL = 83.43           # text encoder at 14 classes
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0,   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
               "auto": 2500.0 / 6}              # f32-equivalent roof of six bf16 products per f32 product
HBM_PEAK_TBPS = 8.0                   # HBM3E nominal (MI355X_MICROARCH.md; ~6.3 achievable with a float4 copy)
HEAD_BATCH = 64                       # configs[1] / configs[3]: 64 videos x 512 features x 512-d per step


def build_net(precision, device, vit_chunk=256, cfg="ucf"):
    import dataclasses
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts
    hc = {"ucf": IW.UCF_HEAD, "xd": dataclasses.replace(IW.XD_HEAD, ncrops=1)}[cfg]      # (xd: the training shape, one crop)
    tab = lookup_prompts(key=cfg)
    toks = torch.tensor(tab["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key=cfg, emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads,
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
    """Oracle (CPU restatement of the reference path) on the host cores, bounded sample; both legs warmed."""
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
        # head leg: one 512-feature tile incl. text encoder, selector, temporal, post-processing (1 warm-up, median of 3)
        feats = torch.randn(1, 1, 512, 512, generator=g) * 0.3
        nc = torch.zeros(512)
        ts = []
        for i in range(4):
            t0 = time.perf_counter()
            sim, sc = O.anomaly_clip_forward_test(sd, hc, feats, nc, eot, 8, 1)
            O.eval_postprocess(sim, sc, 512)
            ts.append(time.perf_counter() - t0)
        t_head = sorted(ts[1:])[1] / 512
    return {"value": round(1.0 / (t_vit + t_head), 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU f32 restatement): ViT-B/16 on {n} frames in batches of 32 "
                      f"({t_vit * 1e3:.1f} ms/frame) + one 512-frame head tile incl. text encoder "
                      f"({t_head * 512 * 1e3:.0f} ms/tile, warm, median of 3)"}


class Timer:
    """barrier + torch.cuda.synchronize() on both sides of the timed region, MAX over ranks."""

    def __init__(self, dist, dev):
        self.dist, self.dev = dist, dev

    def sync(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize()

    def run(self, fn, steps, warmup, on_start=None, on_stop=None):
        for _ in range(warmup):
            fn()
        self.sync()
        if on_start:
            on_start()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.sync()
        dt = time.perf_counter() - t0
        if on_stop:
            on_stop()
        if self.dist is not None:
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt


class Prof:
    """libacx's in-library launch timer (HIP events on the launch stream around every kernel of a kind)."""

    def __init__(self, local_rank):
        from anomalyclip_amd import _lib as L
        self.L, self.h, self.lib = L, L.ctx(local_rank), L.lib()

    def start(self):
        self.lib.acx_prof_enable(self.h, 1)

    def start_gemm_only(self):
        """event pairs around the dominant kernel's launches only: what the timed headline region carries"""
        self.lib.acx_prof_enable(self.h, 2)

    def stop(self):
        self.lib.acx_prof_enable(self.h, 0)

    def collect(self):
        gf = ctypes.c_double(0.0)
        self.L.check(self.lib.acx_prof_gemm_flops(self.h, ctypes.byref(gf)), self.h)
        counts = (ctypes.c_int32 * 4)()
        tot = (ctypes.c_double * 4)()
        self.L.check(self.lib.acx_prof_collect(self.h, counts, tot), self.h)
        return gf.value, list(counts), list(tot)

    def gemm_tn(self):
        """(flops, summed launch ms, launches) of the acx_gemm_tn launches the last collect() consumed."""
        gf, ms, n = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int32(0)
        self.L.check(self.lib.acx_prof_gemm_tn(self.h, ctypes.byref(gf), ctypes.byref(ms), ctypes.byref(n)), self.h)
        return gf.value, ms.value, n.value


def head_batch(B_global, world, rank, dev, step_seed=1, num_classes=14, normal_id=7):
    """UCF-shaped synthetic batch: B videos x 512 x 512-d, first half abnormal (the abnormal classes cycling),
    second half normal; this rank's abnormal/normal-balanced shard as the datamodule's (nbatch, abatch) pair."""
    from anomalyclip_amd import parallel
    g = torch.Generator().manual_seed(step_seed)
    feats = torch.randn(B_global, 1, 512, 512, generator=g) * 0.3
    na = num_classes - 1
    labels = torch.tensor([i % na + (1 if i % na >= normal_id else 0) for i in range(B_global // 2)] + [normal_id] * (B_global // 2))
    idx = parallel.shard_videos(B_global, world, rank)
    f_loc, l_loc = feats[idx].to(dev), labels[idx].to(dev)
    h = len(idx) // 2
    return ((f_loc[h:], l_loc[h:]), (f_loc[:h], l_loc[:h])), idx


def head_legs(net, dev, dist, rank, world, local_rank, steps, warmup, timer):
    """configs[1] (one GPU's worth of head work) and configs[3] (data-parallel training over RCCL)."""
    from anomalyclip_amd import parallel
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.components.loss import ComputeLoss
    crit = ComputeLoss(7, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32)
    net.load_from_features = True
    mod = AnomalyCLIPModule(net, None, None, crit, num_classes=14, solver={"lr": 1e-5}).to(dev)
    mod.ncentroid = torch.zeros(512, device=dev)
    opt = mod.configure_optimizers()["optimizer"]
    prof = Prof(local_rank)
    out = {}

    import contextlib

    @contextlib.contextmanager
    def eager_path():
        """kernel-breakdown passes run the eager path (same kernels; the timed passes replay the graphs)"""
        keep = (net.text_graph, net.temporal_model.graph, getattr(net, "step_graph", True))
        net.text_graph, net.temporal_model.graph, net.step_graph = False, False, False
        try:
            yield
        finally:
            net.text_graph, net.temporal_model.graph, net.step_graph = keep

    def make_train(B_global, w=world, r=rank):
        batch, idx = head_batch(B_global, w, r, dev)
        step_i = [0]

        def step():
            # host mask RNG (selector_model.py:101-117): the GLOBAL batch's mask from a per-step seed, this rank's rows
            torch.manual_seed(step_i[0])
            step_i[0] += 1
            m_top, m_bot = type(net.selector_model).generate_mask(net.selector_model, B_global)
            net.selector_model.generate_mask = lambda b, mt=m_top[idx], mb=m_bot[idx]: (mt, mb)
            mod.train_batch(batch, opt)
        return step, batch

    net.train()
    # train_batch's default is the WHOLE-STEP graph (components/step_graph.py: two linear launch chains replayed from HIP
    # graphs, text tower pipelined across steps, AdamW inside; bit-identical to the eager path, tests/test_gpu_train.py).
    # It is captured inside the first training step and agreed on across ranks there (a rank that cannot capture sends
    # everybody to the autograd path, whose text tower / temporal model are then replayed as separate graphs).
    net.step_graph = True
    net.text_graph = True
    net.temporal_model.graph = True
    tnote = None
    try:
        trial, _ = make_train(HEAD_BATCH)
        trial()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        net.step_graph = False
        net.text_graph = False
        net.temporal_model.graph = False
        tnote = f"{type(e).__name__}: {e}"[:200]
    sgs = mod.__dict__.get("_step_graphs", {})
    out["train_graphs"] = {"whole_step": {"enabled": bool(net.step_graph) and any(v is not None for v in sgs.values()),
                                          "note": getattr(mod, "step_graph_error", None) or tnote},
                           "fallback_text_graph": bool(net.text_graph), "fallback_temporal_graph": bool(net.temporal_model.graph)}
    # ---- configs[1]: one GPU, B = 64 (rank-local copy of the global batch when N > 1: not a scaling leg)
    if world == 1:
        step, batch = make_train(HEAD_BATCH)
        dt = timer.run(step, steps, warmup)
        # kernel breakdown / GEMM roofline from a separate profiled pass (HIP-event pairs around ~400 launches per step
        # cost ~3 ms per step and must not sit inside the features/s measurement)
        psteps = 3
        with eager_path():                       # graph replays are opaque to the library's per-launch event pairs
            timer.run(step, 1, 0)
            timer.run(step, psteps, 0, prof.start, prof.stop)
        gf, counts, tot = prof.collect()
        tn_gf, tn_ms, tn_n = prof.gemm_tn()                      # the weight-gradient launches (acx_gemm_tn) of the profiled steps
        gf, counts, tot = gf * steps / psteps, [c * steps // psteps for c in counts], [t * steps / psteps for t in tot]
        feats = HEAD_BATCH * 512
        gemm_ms, n_gemm = tot[0], counts[0]
        out["head"] = {
            "workload": "configs[1]: 64 videos x 512 x 512-d f32, UCF head (text encoder + selector + axial temporal + "
                        "7-term loss), one GPU",
            "train_features_per_s": round(feats * steps / dt, 1), "train_ms_per_step": round(dt / steps * 1e3, 3),
            "train_gemm": {"tflops": round(gf / 1e9 / gemm_ms, 2) if gemm_ms > 0 else None,
                           "frac_of_f32_mfma_peak": round(gf / 1e9 / gemm_ms / PEAK_TFLOPS["f32"], 4) if gemm_ms > 0 else None,
                           "launches_per_step": n_gemm // steps, "ms_per_step": round(gemm_ms / steps, 3),
                           "executed_gflop_per_step": round(gf / 1e9 / steps, 1),
                           "gemm_tn": {"tflops": round(tn_gf / 1e9 / tn_ms, 2) if tn_ms > 0 else None,
                                       "frac_of_f32_mfma_peak": round(tn_gf / 1e9 / tn_ms / PEAK_TFLOPS["f32"], 4) if tn_ms > 0 else None,
                                       "launches_per_step": tn_n // psteps, "ms_per_step": round(tn_ms / psteps, 3),
                                       "gflop_per_step": round(tn_gf / 1e9 / psteps, 1)},
                           "gemm_nt": {"tflops": round((gf / steps - tn_gf / psteps) / 1e9 / (gemm_ms / steps - tn_ms / psteps), 2)
                                       if gemm_ms / steps > tn_ms / psteps else None,
                                       "ms_per_step": round(gemm_ms / steps - tn_ms / psteps, 3)}},
            "train_kernel_ms_per_step": {"gemm": round(tot[0] / steps, 3), "attention": round(tot[1] / steps, 3),
                                         "norm_rows": round(tot[2] / steps, 3), "other": round(tot[3] / steps, 3)},
        }
        x = torch.cat((batch[1][0], batch[0][0]), 0).view(-1, 1, 512, 512)

        def fwd():
            with torch.no_grad():
                net.eval()
                net(x, None, mod.ncentroid, 1, True)
                net.train()
        dt = timer.run(fwd, steps, warmup)
        out["head"]["fwd_features_per_s"] = round(feats * steps / dt, 1)
        out["head"]["fwd_ms_per_step"] = round(dt / steps * 1e3, 3)
        out["head"]["xd_train"] = xd_train_leg(dev, timer, steps, warmup)

    # ---- configs[3]: data-parallel training (N = 1 gives T1 of both curves)
    dp = {"workload": "configs[3]: UCF-shaped 512-d feature sequences, DP training step = forward + 7-term loss + backward "
                      "(bucketed async gradient all-reduce over RCCL, SyncBN statistics, class-parallel text encoder) + AdamW",
          "n_gpus": world, "grad_bytes": None}
    t1 = None
    if world > 1:
        # T1 inside the same N-rank run: every rank steps the whole 64-video batch alone (no collective, no sharding)
        with parallel.local_only():
            step, _ = make_train(HEAD_BATCH, 1, 0)
            t1 = timer.run(step, steps, warmup) / steps
        dp["t1_ms_per_step_same_run"] = round(t1 * 1e3, 3)
    for mode, B_global in (("strong", HEAD_BATCH), ("weak", HEAD_BATCH * world)):
        if (B_global // 2) % world:
            dp[mode] = None
            continue
        step, _ = make_train(B_global)
        dt = timer.run(step, steps, warmup)
        ent = {"global_batch_videos": B_global, "videos_per_gpu": B_global // world,
               "ms_per_step": round(dt / steps * 1e3, 3), "features_per_s": round(B_global * 512 * steps / dt, 1),
               # whether THIS leg ran on the whole-step graph (one graph set per batch shape) or fell back to the autograd path
               "whole_step_graph": bool(net.step_graph) and getattr(mod, "step_graph_error", None) is None and
                                   any(v is not None for v in mod.__dict__.get("_step_graphs", {}).values()),
               "step_graph_error": getattr(mod, "step_graph_error", None)}
        if t1 is not None:                            # strong: T1 / (N TN); weak: T1 / TN (same per-GPU work as T1)
            ent["efficiency_vs_t1_same_run"] = round(t1 / ((world if mode == "strong" else 1) * dt / steps), 4)
        # per-rank kernel-time breakdown of the same step (separate short pass: the HIP-event pairs around ~700 launches
        # per step would otherwise sit inside the timed region above); rank 0's numbers
        with eager_path():
            timer.run(step, 1, 0)
            timer.run(step, 2, 0, prof.start, prof.stop)
        _, counts, tot = prof.collect()
        ent["rank0_kernel_ms_per_step"] = {"gemm": round(tot[0] / 2, 3), "attention": round(tot[1] / 2, 3),
                                           "norm_rows": round(tot[2] / 2, 3), "other": round(tot[3] / 2, 3),
                                           "launches": sum(counts) // 2}
        dp[mode] = ent
    if mod._buckets is not None:
        flat = mod._buckets.flat
        dp["grad_bytes"] = flat.numel() * 4
        if world > 1:
            def ar():
                dist.all_reduce(flat)
            dt = timer.run(ar, 20, 3)
            dp["allreduce_grad_buffer_ms"] = round(dt / 20 * 1e3, 4)
            dp["allreduce_bus_GBps"] = round(2 * (world - 1) / world * flat.numel() * 4 / (dt / 20) / 1e9, 1)
            mod._buckets.zero()
    out["dp_train"] = dp
    net.load_from_features = False
    net.eval()
    return out


def xd_train_leg(dev, timer, steps, warmup):
    """The XD-Violence head's TRAINING shape (configs/model/anomaly_clip_xdviolence.yaml: E = 128, 7 classes; one crop), 64 videos x 512
    x 512-d per step: round 6 put its convolutions and weight gradients on the bf16 x 6 kernels (256 x 128 / 128 x 256 tile
    geometries); the f32 MFMA kernels beside it.  Only the head: the frozen ViT is not built."""
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.components.loss import ComputeLoss
    ent = {"workload": "XD-Violence head (E = 128, 7 classes, one crop): 64 videos x 512 x 512-d, training step = forward + loss + backward + AdamW"}
    for precision in ("auto", "f32"):
        try:
            net, _, _, hc = build_net(precision, dev, cfg="xd")
            net.load_from_features = True
            crit = ComputeLoss(hc.normal_id, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32)
            mod = AnomalyCLIPModule(net, None, None, crit, num_classes=hc.num_classes, solver={"lr": 1e-5}).to(dev)
            mod.ncentroid = torch.zeros(512, device=dev)
            opt = mod.configure_optimizers()["optimizer"]
            batch, idx = head_batch(HEAD_BATCH, 1, 0, dev, num_classes=hc.num_classes, normal_id=hc.normal_id)
            net.train()
            step_i = [0]

            def step():
                torch.manual_seed(step_i[0])
                step_i[0] += 1
                m_top, m_bot = type(net.selector_model).generate_mask(net.selector_model, HEAD_BATCH)
                net.selector_model.generate_mask = lambda b, mt=m_top[idx], mb=m_bot[idx]: (mt, mb)
                mod.train_batch(batch, opt)
            dt = timer.run(step, steps, warmup)
            ent[precision] = {"train_ms_per_step": round(dt / steps * 1e3, 3), "train_features_per_s": round(HEAD_BATCH * 512 * steps / dt, 1),
                              "x6_convs": bool(net.temporal_model.x6_convs()), "loss": float(mod.last_losses[0].detach()),
                              "whole_step_graph": getattr(mod, "step_graph_error", None) is None}
            del mod, net, opt
        except Exception as e:  # noqa: BLE001
            ent[precision] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return ent


def _event_time(fn, iters, warm=3):
    """average seconds per call of `fn`, HIP events on the current stream (where libacx launches)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def peaks_measured(dev, local_rank):
    """measured denominators next to the nominal ones -- a register-only MFMA loop on every SIMD (f32 and
    bf16) and a 16-byte-per-lane stream copy over 1 GiB (4x the 256 MiB Infinity Cache), both timed with HIP events."""
    from anomalyclip_amd import _lib as L
    lib, h = L.lib(), L.ctx(local_rank)
    st = torch.cuda.current_stream().cuda_stream
    sink = torch.zeros(4, device=dev)
    out = {}
    for name, bf16, nominal in (("mfma_f32_tflops", 0, PEAK_TFLOPS["f32"]), ("mfma_bf16_tflops", 1, PEAK_TFLOPS["bf16"])):
        best = 0.0
        for wps in (1, 2, 4):
            fl = ctypes.c_double(0.0)
            iters = 20000 if bf16 else 4000                        # ~1-2 ms per launch

            def run(wps=wps, iters=iters, fl=fl, bf16=bf16):
                L.check(lib.acx_probe_mfma(h, bf16, iters, wps, sink.data_ptr(), ctypes.byref(fl), st), h)
            dt = _event_time(run, 5, 2)
            best = max(best, fl.value / dt / 1e12)
        out[name] = {"measured": round(best, 1), "nominal": nominal, "ratio": round(best / nominal, 4)}
    # the same bf16 loop with RANDOM operands (four different fragment pairs per lane), sustained for ~0.2 s: what the chip's power
    # limit leaves of the issue rate when the operands toggle like data -- the constant-operand loop above runs at a clock real
    # operands do not sustain.  One wave per SIMD (the plane-reuse kernel's occupancy) and two.
    rnd = {}
    for wps in (1, 2):
        fl = ctypes.c_double(0.0)

        def run_r(wps=wps, fl=fl):
            L.check(lib.acx_probe_mfma(h, 2, 200000, wps, sink.data_ptr(), ctypes.byref(fl), st), h)
        dt = _event_time(run_r, 12 // wps, 2)
        rnd[f"waves_per_simd_{wps}"] = round(fl.value / dt / 1e12, 1)
    best_r = max(rnd.values())
    out["mfma_bf16_random_operands_tflops"] = {"measured": best_r, "by_occupancy": rnd, "nominal": PEAK_TFLOPS["bf16"],
                                               "ratio": round(best_r / PEAK_TFLOPS["bf16"], 4),
                                               "note": "register-only v_mfma_f32_32x32x16_bf16 loop, random bf16 operands, ~0.2 s sustained: "
                                                       "the matrix pipe's rate at the power limit with toggling operands, no memory traffic"}
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)

    def cp():
        L.check(lib.acx_probe_copy(h, src.data_ptr(), dst.data_ptr(), n, st), h)
    dt = _event_time(cp, 10, 2)
    out["hbm_copy_TBps"] = {"measured": round(2 * n / dt / 1e12, 3), "nominal": HBM_PEAK_TBPS,
                            "ratio": round(2 * n / dt / 1e12 / HBM_PEAK_TBPS, 4),
                            "note": "read + write bytes of a 1 GiB float4 grid-stride copy"}
    # write-only stream (acx_fill_f32 over the same GiB): the roof of the write-heavy kernels (frame preprocessing writes
    # 2.6x what it reads) is not the copy rate
    from anomalyclip_amd import ops as _ops
    dstf = dst.view(torch.float32)
    dtf = _event_time(lambda: _ops.fill_(dstf, 1.0), 10, 2)
    out["hbm_fill_TBps"] = {"measured": round(n / dtf / 1e12, 3), "note": "bytes written by a 1 GiB fill (write-only stream)"}
    del src, dst            # stays in torch's cache: torch.cuda.empty_cache() here, with the training legs' HIP graphs alive, made
    return out              # every later allocation of the process a fresh hipMalloc (configs[4] head leg: 1.7 -> 12.7 ms per step)


def hbm_kernel_legs(dev, copy_TBps, fill_TBps=None):
    """The HBM-bound kernels of the path at their benchmark shapes (UCF, B = 64; ViT LayerNorm at 512 frames): algorithmic
    bytes / HIP-event time, as a fraction of the nominal 8 TB/s and of the copy rate measured on this box.  Inputs
    smaller than the 256 MiB Infinity Cache are ROTATED over >= 512 MB of distinct buffers so that a repeat does not read
    them from that cache."""
    from anomalyclip_amd import _lib as L
    from anomalyclip_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    rows, D, C1, E = HEAD_BATCH * 512, 512, 13, 256
    out = {}

    def entry(name, nbytes, dt, note=None):
        tb = nbytes / dt / 1e12
        e = {"bytes": int(nbytes), "us": round(dt * 1e6, 2), "GBps": round(tb * 1e3, 1), "frac_of_8TBps": round(tb / HBM_PEAK_TBPS, 4),
             "frac_of_measured_copy": round(tb / copy_TBps, 4) if copy_TBps else None}
        if note:
            e["note"] = note
        out[name] = e

    def ring(n, *shape):
        return [torch.randn(*shape, generator=g, device=dev) * 0.3 for _ in range(n)]
    nc = torch.zeros(D, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(C1, D, generator=g, device=dev), dim=1)
    xs = ring(8, rows, D)                                                    # 8 x 67 MB
    it = [0]

    def nxt(lst):
        it[0] += 1
        return lst[it[0] % len(lst)]
    entry("selector_project", rows * D * 4 + rows * C1 * 4, _event_time(lambda: ops.selector_project(nxt(xs), nc, dirs), 24),
          "x (32768, 512) -> raw (32768, 13): f32-MFMA skinny GEMM")
    entry("selector_project_stats", rows * D * 4 + rows * C1 * 4,
          _event_time(lambda: ops.selector_project_stats(nxt(xs), nc, dirs), 24), "the same + BatchNorm batch statistics (2 launches; finishing them in the last workgroup to arrive measured slower: 23.5 vs 21.2 us)")
    # the floor of these one-shot launches: a kernel that only READS the same 67 MB (ramp-up and tail included)
    lib, h, st = L.lib(), L.ctx(torch.device(dev).index or 0), torch.cuda.current_stream().cuda_stream
    sink = torch.zeros(4, device=dev)
    read_s = _event_time(lambda: L.check(lib.acx_probe_read(h, nxt(xs).data_ptr(), rows * D * 4, sink.data_ptr(), st), h), 24)
    for k_ in ("selector_project", "selector_project_stats"):
        out[k_]["read_floor_us"] = round(read_s * 1e6, 2)
        out[k_]["frac_of_read_floor"] = round(read_s * 1e6 / out[k_]["us"], 4)
    acc = torch.zeros(D, device=dev)
    entry("colsum_ncentroid", rows * D * 4, _event_time(lambda: ops.colsum_(acc, nxt(xs)), 24))
    del xs
    lw, lb = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    xv = torch.randn(512 * 197, 768, generator=g, device=dev)
    entry("layernorm_vit", 2 * xv.numel() * 4, _event_time(lambda: ops.layernorm(xv, lw, lb), 10), "(100864, 768) in + out")
    del xv
    # row (f2): uint8 HWC frames -> bicubic resize 224 + centre crop + /255 + normalise -> the ViT's f32 input (one fused kernel)
    from anomalyclip_amd.preprocess import preprocess_frames
    fr = [torch.randint(0, 256, (FRAMES_PER_CLIP, 240, 320, 3), generator=g, device=dev, dtype=torch.uint8) for _ in range(5)]
    entry("preprocess_frames", FRAMES_PER_CLIP * (240 * 320 * 3 + 3 * 224 * 224 * 4),
          _event_time(lambda: preprocess_frames(nxt(fr)), 10),
          "512 frames 240x320x3 u8 read + (512, 3, 224, 224) f32 written; Pillow-exact two-pass bicubic, 8-bit intermediate in LDS")
    if fill_TBps:
        wb, rb = FRAMES_PER_CLIP * 3 * 224 * 224 * 4, FRAMES_PER_CLIP * 240 * 320 * 3
        floor_us = (wb / (fill_TBps * 1e12) + rb / (copy_TBps * 1e12)) * 1e6
        out["preprocess_frames"]["frac_of_write_read_floor"] = round(floor_us / out["preprocess_frames"]["us"], 4)
        out["preprocess_frames"]["write_read_floor_us"] = round(floor_us, 1)
    del fr
    x1s, x2s = ring(8, rows, E), ring(8, rows, E)
    cw, cb = torch.randn(1, E, generator=g, device=dev) * 0.1, torch.zeros(1, device=dev)
    l2w, l2b = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    entry("cls_head", 2 * rows * E * 4 + rows * 4, _event_time(lambda: ops.cls_head(nxt(x1s), nxt(x2s), l2w, l2b, cw, cb, 32, 16, 0), 24))
    del x1s, x2s
    raws = ring(64, rows, C1)
    mean, var = torch.zeros(C1, device=dev), torch.ones(C1, device=dev)
    entry("selector_bn", 2 * rows * C1 * 4, _event_time(lambda: ops.selector_bn(nxt(raws), mean, var), 24),
          "1.7 MB in + out: launch-latency bound")
    scs = [torch.rand(rows, generator=g, device=dev) for _ in range(8)]
    entry("class_probs", (2 * rows * C1 + rows) * 4, _event_time(lambda: ops.class_probs(nxt(raws), nxt(scs)), 24),
          "a11 eval post-processing softmax(similarity) * score: 3.5 MB, launch-latency bound")
    del scs
    dls = ring(8, rows, C1)
    entry("bn_bwd_stats", 2 * rows * C1 * 4, _event_time(lambda: ops.bn_bwd_stats(nxt(raws), nxt(dls)), 24), "2 launches, 3.4 MB")
    # loss: (B*512, 13) logits + top-k gather + scores: one fused forward+backward pass
    B = HEAD_BATCH
    sim, simk = raws[0], torch.randn(B * 3 * 16, C1, generator=g, device=dev)
    sc = torch.rand(rows, generator=g, device=dev)
    labels = torch.tensor([i % 13 + (1 if i % 13 >= 7 else 0) for i in range(B // 2)] + [7] * (B // 2), device=dev)
    idx = torch.stack([torch.randperm(32, generator=g, device=dev)[:3] for _ in range(B // 2)])
    lam = (1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3)
    entry("mil_loss", (2 * rows * C1 + 2 * simk.numel() + 2 * rows) * 4,
          _event_time(lambda: ops.mil_loss(sim, simk, labels, sc, idx, idx, idx, 32, 16, 3, 7, lam), 24),
          "reads logits/top-k/scores, writes their gradients; one launch (last-arriver reduction)")
    # the whole-step graph's two fused middle launches (configs[1] batch): selector tail and loss + BatchNorm backward sums
    mt_ = torch.ones(B, 32, device=dev)
    lab_ = labels
    entry("selector_tail", 2 * rows * C1 * 4,
          _event_time(lambda: ops.selector_tail(nxt(raws), lab_, mt_, mt_, 32, 16, 7, 3, 3, 1e-5, stats=(mean, var, var)), 24),
          "BatchNorm + picks + top-k gather of a step in ONE launch (one workgroup per video: 64 workgroups, latency bound); "
          "replaces selector_bn + bn_running_update + select_idx + gather_segments")
    entry("mil_loss_bn", (3 * rows * C1 + simk.numel() + 2 * rows) * 4,
          _event_time(lambda: ops.mil_loss_bn(sim, simk, lab_, sc, idx, idx, idx, 32, 16, 3, 7, lam), 24),
          "loss + gradients + scatter of the top-k gradient + BatchNorm backward sums in ONE launch; replaces mil_loss + axpby + "
          "scatter_segments + bn_bwd_stats (five launches)")
    del raws, dls
    # AdamW over the UCF head's trainable set: one multi-tensor launch, 16 B read + 12 B written per value
    n_par = 10_430_466
    sizes = [n_par // 34] * 33 + [n_par - 33 * (n_par // 34)]
    # FOUR independent (p, g, m, v) sets, used in rotation: 4 x 292 MB = 1.17 GB touched between two visits of the same set,
    # more than four times the 256 MiB Infinity Cache -- an HBM figure, not a cache-assisted one
    rings = []
    for _ in range(4):
        st_ = [[torch.randn(n, generator=g, device=dev) * 0.02 for n in sizes] for _ in range(4)]      # p, g, m, v
        st_[3] = [t.abs() for t in st_[3]]
        rings.append(st_)
    stp = [1]

    def adam():
        stp[0] += 1
        p_, g_, m_, v_ = rings[stp[0] % 4]
        ops.adamw_multi_(p_, g_, m_, v_, [1e-5] * len(sizes), [0.2] * len(sizes), 0.9, 0.999, 1e-8, stp[0])
    entry("adamw_multi", n_par * 28, _event_time(adam, 12, 4),
          "10.43 M parameters in 34 tensors, ONE launch; four parameter sets in rotation (1.17 GB working set)")
    return out


def gemm_error_vs_fp64(dev):
    """One ViT-shaped product (4096 x 768 x 768 + bias, a massive-activation column in A) on both f32-result paths against
    torch's fp64 matmul: max |error| / max |reference|, and / sum_k |a||w| element-wise."""
    from anomalyclip_amd import ops
    g = torch.Generator(device=dev).manual_seed(1)
    M, N, K = 4096, 768, 768
    a = torch.randn(M, K, generator=g, device=dev) * 1.5 + 0.3
    a[:, 5] *= 40.0
    w = torch.randn(N, K, generator=g, device=dev) * 0.05
    b = torch.randn(N, generator=g, device=dev)
    ref = a.double() @ w.double().t() + b.double()
    scale = a.double().abs() @ w.double().abs().t()
    y32 = ops.gemm(a, w, bias=b)
    a3_, w3_ = ops.split_bf16x3(a), ops.split_bf16x3(w)
    y6 = ops.gemm_x6(a3_, w3_, bias=b)
    y3 = ops.gemm_x6(a3_, w3_, bias=b, pairs=3)        # the opt-in three-product mode (precision "bf16x3"): not f32-accurate
    yf = ops.gemm_x6(ops.split_f16x2(a), ops.split_f16x2(w, scale=1024.0), bias=b, pairs=3, out_scale=1.0 / 1024.0)   # opt-in "f16x3"
    out = {"shape": [M, N, K]}
    for name, y in (("f32_mfma", y32), ("bf16x6", y6), ("f16x3_opt_in", yf), ("bf16x3_opt_in", y3)):
        e = (y.double() - ref).abs()
        out[name] = {"max_err_over_max_ref": float(e.max() / ref.abs().max()), "max_err_over_sum_abs_products": float((e / scale).max())}
    return out


def live_pmc_traffic(vit_chunk, precision="auto", timeout_s=150):
    """HBM traffic of the dominant kernel, measured NOW: two separate `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE,
    WRITE_SIZE: they do not fit one pass) over this same script's headline step, exactly as
    MI355X_MICROARCH.md's HBM section prescribes -- FETCH_SIZE is in KB and under-reports wide coalesced reads by 2x on
    gfx950 (x 2), WRITE_SIZE in KB.  Returns (bytes per GEMM launch, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    per = {}
    tmp = tempfile.mkdtemp(prefix="acx_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs",
                   "--no-live-pmc", "--vit-chunk", str(vit_chunk), "--precision", precision]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "p_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            tot, n = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"]
                if row["Counter_Name"] == ctr and "gemm_" in name and "reduce" not in name:
                    tot += float(row["Counter_Value"])
                    n += 1
            if n == 0:
                return None, "no GEMM dispatches in the counter table"
            per[ctr] = tot / n
        b = per["FETCH_SIZE"] * 1024 * 2 + per["WRITE_SIZE"] * 1024
        return int(round(b)), (f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                               f"(1 warm-up + 1 step each), FETCH_SIZE x 2 (gfx950 correction); read "
                               f"{per['FETCH_SIZE'] * 2048 / 1e6:.0f} MB + write {per['WRITE_SIZE'] * 1024 / 1e6:.0f} MB per launch")
    except Exception as e:  # noqa: BLE001
        return None, f"live PMC failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config4_leg(dev, timer, prof, world, steps):
    """BASELINE.json configs[4]: XD-Violence-shaped long segments in bf16 (NOT the parity path): the XD head (C = 7, E = 128)
    on (1, 5 crops, 512 * 16, 512) features with bf16-MFMA GEMMs / implicit-GEMM convolutions, and 5-crop x 32-frame windows
    through the bf16 ViT, four windows (640 frames) per ViT launch -- the reference encodes all frames of a batch in one
    image_encoder call (anomaly_clip.py:118-123, 156-161); `frames_160` keeps the one-window-per-launch figure of rounds 1-2.
    Weak over ranks (independent videos / windows)."""
    from anomalyclip_amd import init_weights as IW
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts
    hc = IW.XD_HEAD
    toks = torch.tensor(lookup_prompts(key="xd")["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key="xd", emb_size=hc.emb_size, depth=hc.depth, heads=hc.heads, dim_heads=None,
                      num_segments=32, seg_length=16, concat_features=False, normal_id=hc.normal_id, stride=1,
                      load_from_features=True, select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=hc.ncrops,
                      num_topk=3, num_bottomk=3, precision="bf16", vit_chunk=640)
    net.load_state_dict(IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, seed=0), strict=True)
    net = net.to(dev).eval()
    S, windows = 16, 4
    g = torch.Generator(device=dev).manual_seed(3)
    feats = torch.randn(1, hc.ncrops, 512 * S, 512, generator=g, device=dev) * 0.3
    nc = torch.zeros(512, device=dev)

    def head():
        with torch.no_grad():
            net(feats, None, nc, S, True)
    net.cache_text_features = False                 # the reference's per-video text tower (anomaly_clip.py:136)
    dt_unc = timer.run(head, steps, 2)
    timer.run(head, 2, 0, prof.start, prof.stop)
    gf_u, _, tot_u = prof.collect()
    net.cache_text_features = True                  # default: frozen prompts in evaluation, text features kept across videos
    dt = timer.run(head, steps, 2)
    timer.run(head, 2, 0, prof.start, prof.stop)
    gf, counts, tot = prof.collect()
    rows = hc.ncrops * 512 * S
    out = {"workload": "configs[4]: XD-Violence shape (C=7, E=128, 5 crops), S=16 tiles per crop, bf16 MFMA / f32 accumulate; "
                       "frames: 5-crop x 32-frame windows, 4 windows = 640 frames per ViT-B/16 launch in bf16 mode",
           "head": {"rows_per_step_per_gpu": rows, "ms_per_step": round(dt / steps * 1e3, 3),
                    "features_per_s": round(rows * steps * world / dt, 1),
                    "gemm_tflops": round(gf / 1e9 / tot[0], 1) if tot[0] else None,
                    "gemm_frac_of_bf16_peak": round(gf / 1e9 / tot[0] / PEAK_TFLOPS["bf16"], 4) if tot[0] else None,
                    "text_features": "kept across videos (evaluation default)",
                    "text_recomputed_per_video": {
                        "ms_per_step": round(dt_unc / steps * 1e3, 3), "features_per_s": round(rows * steps * world / dt_unc, 1),
                        "gemm_tflops": round(gf_u / 1e9 / tot_u[0], 1) if tot_u[0] else None,
                        "gemm_frac_of_bf16_peak": round(gf_u / 1e9 / tot_u[0] / PEAK_TFLOPS["bf16"], 4) if tot_u[0] else None}}}
    frames = torch.randn(160 * windows, 3, 224, 224, generator=g, device=dev)

    def enc():
        with torch.no_grad():
            net.image_encoder(frames)
    k = max(2, steps // 2)
    dt = timer.run(enc, k, 1, prof.start_gemm_only, prof.stop)
    gf, counts, tot = prof.collect()
    out["frames"] = {"frames_per_launch": 160 * windows, "frames_per_s": round(160 * windows * k * world / dt, 1),
                     "gemm_tflops": round(gf / 1e9 / tot[0], 1) if tot[0] else None,
                     "gemm_frac_of_bf16_peak": round(gf / 1e9 / tot[0] / PEAK_TFLOPS["bf16"], 4) if tot[0] else None}
    net.image_encoder.chunk = 160
    dt = timer.run(enc, k, 1, prof.start_gemm_only, prof.stop)
    gf, counts, tot = prof.collect()
    out["frames_160"] = {"frames_per_launch": 160, "frames_per_s": round(160 * windows * k * world / dt, 1),
                         "gemm_tflops": round(gf / 1e9 / tot[0], 1) if tot[0] else None,
                         "gemm_frac_of_bf16_peak": round(gf / 1e9 / tot[0] / PEAK_TFLOPS["bf16"], 4) if tot[0] else None}
    del net, frames, feats
    torch.cuda.empty_cache()
    return out


def self_launch(n):
    """Re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` (rendezvous on
    127.0.0.1, a free port); returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "bf16", "f32x6"],
                    help="auto (default, the headline: f32 results, the large GEMMs as f32-accurate bf16 x 6 products on the bf16 "
                         "matrix cores, the package's default); f32 (the f32 MFMA kernels everywhere: reported as a full leg of the "
                         "default run); bf16 (not a parity path); f32x6 = auto (older name)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the headline step (profiling runs)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not spawn the rocprofv3 counter passes for roofline.traffic")
    ap.add_argument("--vit-chunk", type=int, default=512, help="frames per ViT launch (default: the whole 512-frame clip)")
    ap.add_argument("--leg-limit", type=float, default=1200.0,
                    help="seconds the secondary legs + CPU baseline may take after the headline measurement before rank 0 prints the line "
                         "from the headline alone and exits (0: no limit); a normal run needs ~60 s")
    args = ap.parse_args()
    if args.precision == "f32x6":
        args.precision = "auto"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: become the launcher -- one rank per GPU under torch.distributed.run on this
        # node (the form the driver itself uses); rank 0 of the children prints the single JSON line
        raise SystemExit(self_launch(args.gpus))
    # ACX_BENCH_BACKEND=gloo: functional smoke run of the N > 1 branch on a box with fewer GPUs than ranks (ranks share
    # GPUs, collectives over gloo; the numbers of such a run mean nothing)
    backend = os.environ.get("ACX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); one process "
                         f"per GPU over RCCL needs --gpus <= visible GPUs (ACX_BENCH_BACKEND=gloo shares GPUs for a "
                         f"functional smoke run only)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from anomalyclip_amd import ops

    net, sd, eot, hc = build_net(args.precision, dev, args.vit_chunk)
    g = torch.Generator(device=dev).manual_seed(2 + rank)
    frames = torch.randn(1, FRAMES_PER_CLIP, 3, 224, 224, generator=g, device=dev)      # resident in HBM
    nc = torch.zeros(512, device=dev)
    timer = Timer(dist, dev)
    prof = Prof(local_rank)

    def step():
        with torch.no_grad():
            sim, sc = net(frames, None, nc, 1, True)
            return ops.class_probs(sim, sc), sc

    out_holder = {}

    def step_keep():
        out_holder["o"] = step()

    # ---- library initialisation, outside the W + K protocol: two steps with the per-launch event pairs armed (lazy
    # workspaces, kernel attributes, the profiler's event pool) so that the W warm-up steps warm the chip, not the host
    timer.run(step_keep, 2, 0, prof.start, prof.stop)
    _, counts_all, tot_all = prof.collect()           # per-kind breakdown (attention / norm / other): from these two steps
    # ---- headline: EXACTLY --steps timed steps after --warmup untimed ones.  The timed region carries HIP-event pairs around
    # the dominant kernel's launches (acx_gemm) only: with every launch kind bracketed the ~260 pairs of a step cost 1.9 ms of
    # its 131.5 ms (tools/probes/headline_decomp.py), the ~110 GEMM pairs 0.8 ms
    dt = timer.run(step_keep, args.steps, args.warmup, prof.start_gemm_only, prof.stop)
    gflops_exec, counts, tot = prof.collect()             # GEMM launches of the timed steps only (roofline)
    probs, sc = out_holder["o"]
    assert torch.isfinite(sc).all() and torch.isfinite(probs).all()

    extra = {}
    # ---- the headline is measured: a secondary leg that stops making progress (one unexplained stop of a replayed training step in
    # round 6's test runs) must not take the line with it.  Rank 0 arms a watchdog: after --leg-limit seconds
    # without the final line it prints the line from what IS measured (value, ms_per_step, the GEMM roofline from the timed steps'
    # HIP events; no live traffic pass, no CPU baseline) and ends the process.
    watchdog = None
    if True:
        import threading

        def _give_up(why=None):
            if rank != 0 or _STATE.get("line_printed"):
                # the other ranks leave with rank 0 (a little later, so that its line is out first): a rank left waiting in a collective
                # for a peer that is gone would keep the launcher alive until the collective's own timeout
                os._exit(0)
            _STATE["line_printed"] = True
            n_g, ms_g = counts[0], tot[0]
            ach = (gflops_exec / 1e9 / max(n_g, 1)) / (ms_g / max(n_g, 1)) if ms_g > 0 else 0.0
            line = {"metric": "frames/sec encoded + anomaly-scored (whole node), ViT-B/16 224^2",
                    "value": round(FRAMES_PER_CLIP * args.steps * world / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
                    "config": {"workload": "configs[2]: one 512-frame clip per GPU per step (see DESIGN.md)", "vit_chunk": args.vit_chunk,
                               "precision": args.precision},
                    "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": round(PEAK_TFLOPS[args.precision], 2), "unit": "TFLOP/s",
                                 "frac": round(ach / PEAK_TFLOPS[args.precision], 4), "traffic": None, "launches": int(n_g)},
                    "cpu_baseline": None,
                    "watchdog": (why or f"the secondary legs did not finish within {args.leg_limit} s of the headline measurement") +
                                f": line printed from the headline alone; legs finished so far: {sorted(extra)}"}
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
            os._exit(0)

        # (an exception that escapes main() after this point -- a collective failing because a peer is gone -- ends the same way: __main__)
        _STATE["give_up"] = _give_up
        if args.leg_limit > 0:
            if world > 1:                    # the N > 1 legs (data-parallel training over RCCL) add ~20 s to a normal run: a shorter leash
                args.leg_limit = min(args.leg_limit, 420.0)
            watchdog = threading.Timer(args.leg_limit + (0.0 if rank == 0 else 5.0), _give_up)
            watchdog.daemon = True
            watchdog.start()
    if not args.no_extra_legs:
        k = max(2, min(args.steps, 5))
        # encode-only
        flat_frames = frames.view(-1, 3, 224, 224)

        def enc():
            with torch.no_grad():
                net.image_encoder(flat_frames)
        dte = timer.run(enc, k, 1)
        extra["encode_only"] = {"frames_per_s": round(FRAMES_PER_CLIP * k * world / dte, 2), "ms_per_clip": round(dte / k * 1e3, 3)}
        # the evaluation text-feature cache is ON by default (the prompts are frozen under no_grad: the reference's per-video
        # text tower returns the same tensor every step); the reference's per-step recomputation for comparison
        net.cache_text_features = False
        dtu = timer.run(step_keep, k, 1)
        net.cache_text_features = True
        extra["text_recomputed_every_step"] = {"frames_per_s": round(FRAMES_PER_CLIP * k * world / dtu, 2),
                                               "ms_per_step": round(dtu / k * 1e3, 3),
                                               "note": "cache_text_features = False (anomaly_clip.py:136 semantics); the headline runs the default"}
        if args.vit_chunk != 256:
            net.image_encoder.chunk = 256
            dt256 = timer.run(step_keep, k, 1)
            net.image_encoder.chunk = args.vit_chunk
            extra["vit_chunk_256"] = {"frames_per_s": round(FRAMES_PER_CLIP * k * world / dt256, 2),
                                      "ms_per_step": round(dt256 / k * 1e3, 3),
                                      "note": "config[2]'s literal batch: two 256-frame ViT launches per clip"}
            if args.precision == "auto":
                # the opt-in K split of the partly filled last round of tiles (ACX_OPT_X6_TAIL_SPLIT; off by default: it gives the
                # tail rows of a launch another summation order than the rows before them)
                from anomalyclip_amd import ops as _ops
                try:
                    _ops.set_x6_tail_split(local_rank, True)
                    net.image_encoder.chunk = 256
                    dt256t = timer.run(step_keep, k, 1)
                    net.image_encoder.chunk = args.vit_chunk
                    dt512t = timer.run(step_keep, k, 1)
                finally:
                    _ops.set_x6_tail_split(local_rank, False)
                    net.image_encoder.chunk = args.vit_chunk
                extra["x6_tail_split_opt_in"] = {"frames_per_s": round(FRAMES_PER_CLIP * k * world / dt512t, 2),
                                                 "frames_per_s_vit_chunk_256": round(FRAMES_PER_CLIP * k * world / dt256t, 2),
                                                 "note": "acx_set_option(ACX_OPT_X6_TAIL_SPLIT, 1): not the default, not the headline"}
        if args.precision in ("auto", "f32"):
            # the OTHER f32-result path as a full leg: under the default (auto: the large GEMMs as f32-ACCURATE products on the
            # bf16 matrix cores -- three bf16 planes per operand, six cross products, f32 accumulation: acx_gemm_desc.pairs,
            # acx_gemm_x6.h) the f32 MFMA kernels everywhere, and vice versa; with the distance between the two outputs and a
            # GEMM-level check of both against fp64
            other = "f32" if args.precision == "auto" else "auto"
            leg = "f32_mfma_path" if other == "f32" else "f32_via_bf16x6"
            try:
                ref_probs, ref_sc = (t.clone() for t in out_holder["o"])
                net.image_encoder.precision = other
                timer.run(step_keep, 1, 0)
                p6, s6 = out_holder["o"]
                den_p, den_s = float(ref_probs.abs().max()), float(ref_sc.abs().max())
                dt6 = timer.run(step_keep, k, 1)
                prof.start_gemm_only(); timer.run(step_keep, 1, 0); prof.stop()
                gf6, c6, t6 = prof.collect()
                mult = 6 if other == "auto" else 1
                extra[leg] = {
                    "frames_per_s": round(FRAMES_PER_CLIP * k * world / dt6, 2), "ms_per_step": round(dt6 / k * 1e3, 3),
                    "max_abs_diff_vs_headline_path": {"class_probs": float((p6 - ref_probs).abs().max()), "scores": float((s6 - ref_sc).abs().max()),
                                                      "relative_to_max": [float((p6 - ref_probs).abs().max()) / max(den_p, 1e-30),
                                                                          float((s6 - ref_sc).abs().max()) / max(den_s, 1e-30)]},
                    "roofline": {"bound": "mfma",
                                 "kernel": ("acx_gemm (gemm_f32_p256_kernel / gemm_f32_w8_kernel, v_mfma_f32_32x32x2_f32)" if other == "f32" else
                                            "acx_gemm (gemm_x6_p4_kernel: six v_mfma_f32_32x32x16_bf16 products per f32 product)"),
                                 "launches": c6[0], "gemm_ms_per_step": round(t6[0], 3),
                                 "achieved": round(gf6 / 1e9 / t6[0], 2) if t6[0] else None, "peak": round(PEAK_TFLOPS[other], 2),
                                 "unit": "TFLOP/s (f32-equivalent: 2 M N K per launch)",
                                 "frac": round(gf6 / 1e9 / t6[0] / PEAK_TFLOPS[other], 4) if t6[0] else None,
                                 "executed_mfma_tflops": round(mult * gf6 / 1e9 / t6[0], 1) if t6[0] else None},
                    "note": ("precision 'f32': every GEMM on the f32 MFMA (an fmaf chain per output), the parity path of rounds 1-4" if other == "f32" else
                             "precision 'auto': x = hi + mid + lo in bf16 (exact 24-bit split), products (hi,lo) (mid,mid) (lo,hi) (hi,mid) "
                             "(mid,hi) (hi,hi) on v_mfma_f32_32x32x16_bf16, f32 accumulation")}
            except Exception as e:  # noqa: BLE001
                extra[leg] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                net.image_encoder.precision = args.precision
            if args.precision == "auto":
                # the two opt-in three-product arithmetics of the plane kernels, never the headline:
                #   "f16x3"  TWO fp16 planes per operand (the f32 value to 2^-24), products (lo,hi) (hi,lo) (hi,hi) exact on the fp16 matrix
                #            cores: f32-MFMA-level results (the default's test bounds) for operands inside fp16's range
                #   "bf16x3" the three LEADING products of the default's bf16 split: sixteen significant bits per operand (~1e-5 per product)
                # -- frames/s and the distance from the headline path's outputs
                for mode, leg, desc in (("f16x3", "f16x3_mode_opt_in",
                                         "VisionTransformer(precision = 'f16x3') / ACX_PREC_F16X3: two fp16 planes per operand (hi = fp16(x), lo = fp16(x - hi)), "
                                         "products (lo,hi) (hi,lo) (hi,hi) on v_mfma_f32_32x32x16_f16, f32 accumulation; holds the default's golden bounds "
                                         "(tests/test_gpu_model.py::test_vit_b16_f16x3_mode, test_gpu_kernels.py::test_gemm_f16x3_is_f32_accurate); operands must "
                                         "lie inside fp16's range (|x| < 65504, weights' planes scaled by 2^10): not f32's range, hence opt-in"),
                                        ("bf16x3", "bf16x3_mode_opt_in",
                                         "VisionTransformer(precision = 'bf16x3') / ACX_PREC_F32X3: products (mid,hi) (hi,mid) (hi,hi) of the exact 24-bit bf16 plane "
                                         "split, the lo planes neither written nor read; ViT-B/16 features within 5e-5 of the reference's "
                                         "(tests/test_gpu_model.py::test_vit_b16_three_product_mode), inside BASELINE.json's 1e-3; not f32-accurate")):
                    try:
                        ref_probs, ref_sc = (t.clone() for t in out_holder["o"])
                        net.image_encoder.precision = mode
                        timer.run(step_keep, 1, 0)
                        p3, s3 = out_holder["o"]
                        dt3 = timer.run(step_keep, k, 1)
                        prof.start_gemm_only(); timer.run(step_keep, 1, 0); prof.stop()
                        gf3, c3, t3 = prof.collect()
                        extra[leg] = {
                            "frames_per_s": round(FRAMES_PER_CLIP * k * world / dt3, 2), "ms_per_step": round(dt3 / k * 1e3, 3),
                            "max_abs_diff_vs_headline_path": {"class_probs": float((p3 - ref_probs).abs().max()), "scores": float((s3 - ref_sc).abs().max()),
                                                              "relative_to_max": [float((p3 - ref_probs).abs().max()) / max(float(ref_probs.abs().max()), 1e-30),
                                                                                  float((s3 - ref_sc).abs().max()) / max(float(ref_sc.abs().max()), 1e-30)]},
                            "roofline": {"bound": "mfma", "kernel": "acx_gemm (gemm_x6_p4_kernel<.., X3>: three 16-bit MFMA products per f32 product)",
                                         "launches": c3[0], "gemm_ms_per_step": round(t3[0], 3),
                                         "achieved": round(gf3 / 1e9 / t3[0], 2) if t3[0] else None, "peak": round(PEAK_TFLOPS["bf16"] / 3, 2),
                                         "unit": "TFLOP/s (f32-equivalent: 2 M N K per launch; x 3 = executed 16-bit TFLOP/s)",
                                         "frac": round(3 * gf3 / 1e9 / t3[0] / PEAK_TFLOPS["bf16"], 4) if t3[0] else None},
                            "note": desc}
                        if mode == "f16x3":              # worst-case operand magnitudes PROVEN from the weights against fp16's 65504
                            rep = net.image_encoder.f16x3_range_report()
                            extra[leg]["operand_range_proven_from_weights"] = {
                                "bounds": {k_: round(v_, 2) for k_, v_ in rep["bounds"].items()}, "fp16_max": rep["fp16_max"],
                                "margin": round(rep["margin"], 1), "safe": rep["safe"]}
                    except Exception as e:  # noqa: BLE001
                        extra[leg] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    finally:
                        net.image_encoder.precision = args.precision
            try:
                extra["gemm_error_vs_fp64"] = gemm_error_vs_fp64(dev)
            except Exception as e:  # noqa: BLE001
                extra["gemm_error_vs_fp64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            # the secondary legs must not take the headline line with them (an exception raised on every rank alike --
            # out of memory, an unsupported collective -- is reported in place of the leg's numbers)
            try:
                extra.update(head_legs(net, dev, dist, rank, world, local_rank, max(4, min(args.steps, 10)), 2, timer))
            except Exception as e:  # noqa: BLE001
                extra["head_legs_error"] = f"{type(e).__name__}: {e}"[:300]
            if world == 1:
                try:
                    pk = peaks_measured(dev, local_rank)
                    extra["peaks_measured"] = pk
                    extra["hbm_kernels"] = hbm_kernel_legs(dev, pk["hbm_copy_TBps"]["measured"], pk.get("hbm_fill_TBps", {}).get("measured"))
                except Exception as e:  # noqa: BLE001
                    extra["peaks_measured_error"] = f"{type(e).__name__}: {e}"[:300]
            try:
                extra["config4_xd_bf16"] = config4_leg(dev, timer, prof, world, max(4, min(args.steps, 10)))
            except Exception as e:  # noqa: BLE001
                extra["config4_xd_bf16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # LAST of the legs: it creates two more HIP streams, and the streams of a process share four hardware queues -- created before
        # the training legs, they put the step graph's text stream onto the main stream's queue (head step 10.8 -> 15.8 ms in one refresh
        # run; since then the step graph places its text stream by a measured test, ops.side_stream_beside -- the order stays)
        if args.vit_chunk != 256:
            # opt-in: the clip as two 256-frame half batches on two streams (VisionTransformer.streams = 2): one half's LayerNorm /
            # attention launches run beside the other half's GEMMs.  Not the headline: overlapping launches have no per-launch time
            try:
                net.load_from_features = False          # (the training legs switched the net to feature input and stepped its head)
                net.eval()
                net.image_encoder.streams = 2
                dt2s = timer.run(step_keep, k, 1)
                extra["vit_two_streams"] = {"frames_per_s": round(FRAMES_PER_CLIP * k * world / dt2s, 2), "ms_per_step": round(dt2s / k * 1e3, 3),
                                            "note": "VisionTransformer(streams = 2), opt-in: two 256-frame halves of the clip on two HIP streams with "
                                                    "their own workspaces; same kernels and bits as two 256-frame launches"}
            except Exception as e:  # noqa: BLE001
                extra["vit_two_streams"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                net.image_encoder.streams = 1

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = FRAMES_PER_CLIP * args.steps * world / dt
        # roofline of the dominant kernel (acx_gemm): algorithmic flops per launch / avg launch time.
        # per step the GEMM kernel runs the ViT GEMMs of 512 frames, the text encoder's GEMMs and the
        # head's GEMMs/convs; algorithmic flops = SURVEY 8(d) per-unit figures x units per step.
        # (the headline step runs with the evaluation text-feature cache: no text-encoder launch inside it -- its flops belong to
        # the `text_recomputed_every_step` leg only)
        gemm_gflop_step = GEMM_GFLOP_PER_FRAME * FRAMES_PER_CLIP + HEAD_GFLOP_PER_TILE - 0.025
        n_gemm, ms_gemm = counts[0], tot[0]
        avg_ms = ms_gemm / max(n_gemm, 1)
        # The last ViT layer is evaluated only where its output is consumed (CLS token, clip/model.py:285), so
        # the GEMM kernel EXECUTES fewer flops than the reference's dense formulation: the roofline uses the
        # flops the launches really computed (2*M*N*K summed by libacx), the dense figure is reported beside it.
        gemm_gflop_exec_step = gflops_exec / 1e9 / args.steps
        flop_per_launch = gemm_gflop_exec_step * args.steps / max(n_gemm, 1)  # GFLOP per launch (average)
        achieved = flop_per_launch / avg_ms if avg_ms > 0 else 0.0           # GFLOP/ms == TFLOP/s
        peak = PEAK_TFLOPS[args.precision]
        # HBM traffic of the dominant kernel: PMC counters cannot be read in-process; they are collected by
        # tools/profile_bench.sh (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
        # FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/.
        traffic, pmc_src = None, None
        if world == 1 and args.precision in ("auto", "f32") and not args.no_live_pmc and not args.no_extra_legs:
            torch.cuda.synchronize()
            traffic, pmc_src = live_pmc_traffic(args.vit_chunk, args.precision)
            if traffic is None:
                pmc_src = None
        for tag in (("r06", "r05", "r04", "r03") if traffic is None else ()):
            pmc_file = os.path.join(REPO, "profiles", f"{tag}_bench_{args.precision}_pmc.json")
            if args.precision in ("auto", "f32") and args.vit_chunk == 512 and os.path.exists(pmc_file):
                try:
                    traffic = round(json.load(open(pmc_file))["gemm"]["hbm_bytes_per_launch"])
                    pmc_src = f"profiles/{tag}_bench_{args.precision}_pmc.json (rocprofv3 PMC passes of this command, bytes per launch)"
                    break
                except Exception:
                    traffic = None
        # algorithmic bytes of the step's GEMM launches (f32: 4 B; ViT at 512 frames x 197 tokens, width 768): per layer QKV
        # (A + W + C), out-proj (+ residual), FC, proj (+ residual); patch embed; the head / text GEMMs are < 1 % and left out
        Mv, Wd = FRAMES_PER_CLIP * 197, 768
        if args.precision == "auto":
            # the four large GEMMs of a layer read THREE bf16 planes of A and of W (6 B per element) and write f32 (qkv, out, proj;
            # the latter two also read the f32 residual) or three bf16 planes (c_fc -> c_proj's operand)
            per_layer = (6 * Mv * Wd + 6 * 3 * Wd * Wd + 4 * Mv * 3 * Wd) + (6 * Mv * Wd + 6 * Wd * Wd + 8 * Mv * Wd) \
                + (6 * Mv * Wd + 6 * 4 * Wd * Wd + 6 * Mv * 4 * Wd) + (6 * Mv * 4 * Wd + 6 * 4 * Wd * Wd + 8 * Mv * Wd)
            algo_bytes_step = 11 * per_layer + 4 * (FRAMES_PER_CLIP * 196 * (768 + Wd) + 768 * Wd + 3 * Mv * Wd)
        else:
            bpe = 2 if args.precision == "bf16" else 4
            per_layer = (Mv * Wd + 3 * Wd * Wd + Mv * 3 * Wd) + (Mv * Wd + Wd * Wd + 2 * Mv * Wd) + (Mv * Wd + 4 * Wd * Wd + Mv * 4 * Wd) \
                + (Mv * 4 * Wd + 4 * Wd * Wd + 2 * Mv * Wd)
            algo_bytes_step = bpe * (11 * per_layer + FRAMES_PER_CLIP * 196 * (768 + Wd) + 768 * Wd + 3 * Mv * Wd)
        algo_bytes_per_launch = algo_bytes_step / max(n_gemm / args.steps, 1) if n_gemm else None
        out = {
            "metric": "frames/sec encoded + anomaly-scored (whole node), ViT-B/16 224^2",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
            "world": {"size": world, "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None}, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic 224x224 RGB frames, ViT-B/16 encode + selector + axial temporal head + eval "
                                   "post-processing; step = one 512-frame clip per GPU in ONE ViT launch, UCF-Crime head config, "
                                   "random-init weights; text features cached (frozen prompts): leg text_recomputed_every_step "
                                   "runs the text encoder in every step; leg vit_chunk_256 = the config's literal batch of 256",
                       "frames_per_step_per_gpu": FRAMES_PER_CLIP, "vit_chunk": args.vit_chunk, "precision": args.precision},
            "roofline": {"bound": "mfma", "kernel": {
                             "f32": "acx_gemm (gemm_f32_p256_kernel / gemm_f32_w8_kernel, v_mfma_f32_32x32x2_f32)",
                             "bf16": "acx_gemm (gemm_bf16_p8_kernel / gemm_bf16_dma_kernel / gemm_kernel, v_mfma_f32_32x32x16_bf16)",
                             "auto": "acx_gemm (gemm_x6_p4_kernel, six v_mfma_f32_32x32x16_bf16 products per f32 product: achieved / "
                                     "peak in f32-EQUIVALENT TFLOP/s, peak = 2500 / 6 = 416.7; x 6 = executed bf16 TFLOP/s against 2500)"}[args.precision],
                         "achieved": round(achieved, 2), "peak": round(peak, 2), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "executed_bf16_tflops": round(6 * achieved, 1) if args.precision == "auto" else None,
                         "frac_of_measured_mfma_peak": (round((6 if args.precision == "auto" else 1) * achieved / extra["peaks_measured"][
                             "mfma_f32_tflops" if args.precision == "f32" else "mfma_bf16_tflops"]["measured"], 4)
                             if "peaks_measured" in extra else None),
                         # against the register-only loop with RANDOM operands (the rate the power limit leaves with toggling data)
                         "frac_of_sustained_mfma_rate_random_operands": (
                             round(6 * achieved / extra["peaks_measured"]["mfma_bf16_random_operands_tflops"]["measured"], 4)
                             if args.precision == "auto" and "mfma_bf16_random_operands_tflops" in extra.get("peaks_measured", {}) else None),
                         "traffic": traffic, "traffic_source": pmc_src,
                         # counter bytes / algorithmic bytes of a GEMM launch (A + W read once, C written once; residuals and
                         # biases included): > 1 = re-reads through the fabric
                         "wasted_traffic_ratio": (round(traffic / algo_bytes_per_launch, 3) if traffic and algo_bytes_per_launch else None),
                         "algorithmic_bytes_per_launch": int(algo_bytes_per_launch) if algo_bytes_per_launch else None,
                         "launches": int(n_gemm), "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_gflop_per_launch": round(flop_per_launch, 3),
                         "gemm_gflop_per_step_executed": round(gemm_gflop_exec_step, 1),
                         "gemm_gflop_per_step_dense_reference": round(gemm_gflop_step, 1)},
            # ONE consistent breakdown: every kind from the same two untimed, fully bracketed steps (event pairs around all ~260
            # launches; those steps run ~2 ms longer than the timed ones, so the kinds add up to a bracketed step, not to
            # ms_per_step -- the GEMM time of the TIMED steps is roofline.avg_launch_ms x roofline.launches / steps)
            "gemm_ms_per_timed_step": round(ms_gemm / args.steps, 3),
            "kernel_time_ms_per_step": {"gemm": round(tot_all[0] / 2, 3), "attention": round(tot_all[1] / 2, 3),
                                        "norm_rows": round(tot_all[2] / 2, 3), "other": round(tot_all[3] / 2, 3),
                                        "source": "the two INITIALISATION steps, every launch bracketed by HIP events (they run ~2 ms "
                                                  "longer than a timed step: compare gemm_ms_per_timed_step, not ms_per_step)"},
            "end_to_end_tflops": round((VIT_GFLOP_PER_FRAME * FRAMES_PER_CLIP + HEAD_GFLOP_PER_TILE) * world / ms_per_step, 2),
        }
        out.update(extra)
        # the secondary figures a reader wants next to `value`, lifted to the top level
        out["value_vit_chunk_256"] = extra.get("vit_chunk_256", {}).get("frames_per_s")
        out["value_f32_mfma_path"] = extra.get("f32_mfma_path", {}).get("frames_per_s")
        out["value_vit_two_streams_opt_in"] = extra.get("vit_two_streams", {}).get("frames_per_s")
        out["value_f16x3_mode_opt_in"] = extra.get("f16x3_mode_opt_in", {}).get("frames_per_s")
        out["value_bf16x3_mode_opt_in"] = extra.get("bf16x3_mode_opt_in", {}).get("frames_per_s")
        out["value_text_recomputed_every_step"] = extra.get("text_recomputed_every_step", {}).get("frames_per_s")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, eot, hc)
        if watchdog is not None:
            watchdog.cancel()
        _STATE["line_printed"] = True
        print(json.dumps(out), flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        if "give_up" not in _STATE:          # nothing measured yet: fail loudly
            raise
        import traceback
        traceback.print_exc()
        _STATE["give_up"](f"{type(e).__name__} after the headline measurement ({str(e)[:200]})")
