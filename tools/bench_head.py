#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[1] and [3]): UCF-Crime-shaped synthetic 512-d feature sequences through
the head -- text encoder + selector + axial temporal transformer + loss, forward-only and full training step
(forward, 7-term loss, backward, AdamW) -- features/s = videos * 512 * steps / time.  bench.py runs the same legs
(`head`, `dp_train`) inside the driver's command; this tool adds knobs for development:

    python tools/bench_head.py [--batch 64] [--steps 10] [--warmup 2]
    python -m torch.distributed.run --nproc-per-node N tools/bench_head.py --gpus N [--scaling strong|weak]
    python tools/bench_head.py --emulate-world 8      # ONE GPU: rank 0's share of an 8-rank strong-scaling step
                                                      # (8 videos, 2 of 14 text classes, collectives stubbed out):
                                                      # per-rank compute + launch overhead = the efficiency ceiling

Data parallel: videos shard across ranks (parallel.shard_videos), gradients through parallel.GradBuckets
(bucketed async all-reduce over RCCL), SyncBN statistics inside the selector, text encoder class-parallel.
One JSON line on rank 0."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def stub_collectives(world, net=None):
    """single-process stand-in for torch.distributed inside anomalyclip_amd.parallel (timing of rank 0's share only).
    The class-parallel text exchange (parallel.assemble_rows) is emulated with REAL data: the text features of all
    classes are evaluated once here, untimed, and the stubbed exchange fills the other ranks' class rows from that copy --
    a no-op all-reduce would leave them zero, (T - c)/|T - c| = 0/0 with the zero ncentroid of the benchmark, and every
    kernel downstream would be timed on NaNs."""
    from anomalyclip_amd import parallel
    if net is not None:
        import torch as _t
        from anomalyclip_amd.components import functional as Fn
        with _t.no_grad():
            full, _ = Fn._text_forward_rows(net, net.prompt_learner.ctx, net.text_encoder.text_projection, 0,
                                            net.prompt_learner.n_cls)
        full = full.detach().clone()

        def assemble_rows(local, lo, n):
            out = full.clone()
            out[lo:lo + local.shape[0]] = local
            return out
        parallel.assemble_rows = assemble_rows

        def assemble_rows_into(buf, lo, hi):         # the step graph's static exchange buffer: foreign rows from the copy
            buf[:lo].copy_(full[:lo])
            buf[hi:].copy_(full[hi:])
            return buf
        parallel.assemble_rows_into = assemble_rows_into

    # SyncBN all-gather of the step graph: ONE collective on real hardware -> one broadcast copy here (the list form of the stub
    # below is eight copies)
    parallel.all_gather_into = lambda out, local: out.copy_(local.unsqueeze(0).expand_as(out))

    class _H:
        def wait(self):
            return None

    class _Dist:
        ReduceOp = SimpleNamespace(SUM=0, MAX=1, MIN=2)

        @staticmethod
        def all_reduce(t, op=None, async_op=False):
            return _H() if async_op else None

        @staticmethod
        def all_gather(lst, t):
            for x in lst:
                x.copy_(t)

        @staticmethod
        def get_world_size():
            return world

        @staticmethod
        def get_rank():
            return 0

        @staticmethod
        def is_available():
            return True

        @staticmethod
        def is_initialized():
            return True

    parallel.dist = _Dist
    parallel.is_distributed = lambda: True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch (videos) for strong scaling, per-GPU for weak")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--emulate-world", type=int, default=0)
    ap.add_argument("--no-text-shard", action="store_true", help="replicate the text encoder on every rank (plain DP)")
    ap.add_argument("--text-graph", action="store_true", help="text tower as two replayed HIP graphs on a side stream")
    ap.add_argument("--temporal-graph", action="store_true", help="temporal model forward / backward as two replayed HIP graphs")
    ap.add_argument("--no-step-graph", action="store_true", help="autograd path instead of train_batch's whole-step graph")
    ap.add_argument("--x6-cus", type=int, default=-1, help="ACX_OPT_X6_CUS: CUs the bf16 x 6 kernels may hold (0 = all; -1 = the module's default)")
    ap.add_argument("--x6-tail", action="store_true", help="ACX_OPT_X6_TAIL_SPLIT on (K-split a partly filled last round of tiles)")
    ap.add_argument("--precision", default="auto", choices=["auto", "f32"], help="auto: the convolutions as bf16 x 6 products (default)")
    ap.add_argument("--config", default="ucf", choices=["ucf", "xd"], help="head configuration (xd: E = 128, 7 classes, one crop)")
    ap.add_argument("--extra-streams", type=int, default=0,
                    help="development probe: create (and use once) this many HIP streams BEFORE the step graph exists -- the streams of a "
                         "process share GPU_MAX_HW_QUEUES (default 4) hardware queues; a text stream that lands on the main stream's queue "
                         "serialises the step's two chains")
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    import bench as B
    from anomalyclip_amd import parallel
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.components.loss import ComputeLoss

    net, sd, eot, hc = B.build_net(args.precision, dev, cfg=args.config)
    net.load_from_features = True
    net.text_class_parallel = not args.no_text_shard
    net.text_graph = bool(args.text_graph)
    net.temporal_model.graph = bool(args.temporal_graph)
    net.step_graph = not args.no_step_graph
    crit = ComputeLoss(hc.normal_id, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32)
    mod = AnomalyCLIPModule(net, None, None, crit, num_classes=hc.num_classes, solver={"lr": 1e-5}).to(dev)
    mod.ncentroid = torch.zeros(512, device=dev)
    opt = mod.configure_optimizers()["optimizer"]
    timer = B.Timer(dist, dev)
    prof = B.Prof(local_rank)

    eff_world = world
    if args.emulate_world > 1:
        assert world == 1
        eff_world = args.emulate_world
    B_global = args.batch if args.scaling == "strong" else args.batch * eff_world
    batch, idx = B.head_batch(B_global, eff_world, rank, dev, num_classes=hc.num_classes, normal_id=hc.normal_id)
    if args.emulate_world > 1:
        stub_collectives(eff_world, net)

    extra_streams = []
    for _ in range(args.extra_streams):
        st_ = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st_):
            torch.zeros(16, device=dev).add_(1.0)
        extra_streams.append(st_)
    torch.cuda.synchronize()
    net.train()
    step_i = [0]
    if args.x6_cus >= 0:
        from anomalyclip_amd import ops as ops_
        ops_.set_x6_cus(local_rank, args.x6_cus)
    if args.x6_tail:
        from anomalyclip_amd import ops as ops_
        ops_.set_x6_tail_split(local_rank, True)
    if os.environ.get("ACX_TN_P256_MIN_ROWS"):           # development A/B: 2147483647 keeps the 128 x 128 weight-gradient kernels
        from anomalyclip_amd import _lib as L_
        L_.check(L_.lib().acx_set_option(L_.ctx(local_rank), L_.OPT_TN_P256_MIN_ROWS, int(os.environ["ACX_TN_P256_MIN_ROWS"])),
                 L_.ctx(local_rank))
    loss_trace = []

    def train_step():
        torch.manual_seed(step_i[0])          # host mask RNG: the global batch's mask, this rank's rows
        step_i[0] += 1
        mt, mb = type(net.selector_model).generate_mask(net.selector_model, B_global)
        net.selector_model.generate_mask = lambda b, mt=mt[idx], mb=mb[idx]: (mt, mb)
        mod.train_batch(batch, opt)
        if len(loss_trace) < 4:
            loss_trace.append(mod.last_losses[0].detach().clone())

    x = torch.cat((batch[1][0], batch[0][0]), 0).view(-1, 1, 512, 512)

    def fwd_step():
        with torch.no_grad():
            net.eval()
            net(x, None, mod.ncentroid, 1, True)
            net.train()

    if os.environ.get("ACX_CPROFILE"):
        # where the HOST spends a step (development aid): cumulative-time table of 20 steps to stderr
        import cProfile
        import pstats
        timer.run(train_step, 3, 0)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            train_step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(25)
    dt_train = timer.run(train_step, args.steps, args.warmup)
    keep_sg = net.step_graph
    net.step_graph = False                     # per-launch event pairs see library calls, not graph replays
    timer.run(train_step, 1, 0)
    timer.run(train_step, 2, 0, prof.start, prof.stop)
    net.step_graph = keep_sg
    gf, counts, tot = prof.collect()
    dt_fwd = timer.run(fwd_step, args.steps, args.warmup)
    # host-side cost of issuing one step (no device wait inside): launch-bound when close to ms_per_step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step()
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    if rank == 0:
        feats_per_step = B_global * 512
        print(json.dumps({
            "metric": "features/sec through the head (UCF-Crime shape, 512-d), train step = fwd+loss+bwd+AdamW",
            "train_features_per_s": round(feats_per_step * args.steps / dt_train, 1),
            "train_ms_per_step": round(dt_train / args.steps * 1e3, 3),
            "fwd_features_per_s": round(feats_per_step * args.steps / dt_fwd, 1),
            "fwd_ms_per_step": round(dt_fwd / args.steps * 1e3, 3),
            "host_issue_ms_per_train_step": round(host_ms, 3),
            "kernel_ms_per_train_step": {"gemm": round(tot[0] / 2, 3), "attention": round(tot[1] / 2, 3),
                                         "norm_rows": round(tot[2] / 2, 3), "other": round(tot[3] / 2, 3),
                                         "launches": sum(counts) // 2},
            "train_gemm_tflops": round(gf / 1e9 / tot[0], 2) if tot[0] > 0 else None,
            "n_gpus": world, "emulated_world": args.emulate_world or None, "videos_this_rank": len(idx),
            "text_class_parallel": bool(net.text_class_parallel), "text_graph": bool(net.text_graph), "temporal_graph": bool(net.temporal_model.graph),
            "step_graph": bool(net.step_graph) and any(v is not None for v in mod.__dict__.get("_step_graphs", {}).values()),
            "step_graph_error": getattr(mod, "step_graph_error", None),
            "global_batch_videos": B_global, "scaling": args.scaling, "dtype": "f32",
            "loss": float(mod.last_losses[0].detach()), "first_losses": [round(float(v), 6) for v in loss_trace], "data": "synthetic"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
