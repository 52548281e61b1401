#!/usr/bin/env python
"""Secondary benchmark: UCF-Crime-shaped synthetic 512-d feature sequences through
the head -- text encoder + selector + axial temporal transformer + loss, forward-only and full training step
(forward, 7-term loss, backward, AdamW) -- features/s = videos * 512 * steps / time.

    python tools/bench_head.py [--batch 64] [--steps 10] [--warmup 2]
    python -m torch.distributed.run --nproc-per-node N tools/bench_head.py --gpus N [--scaling strong|weak]

Data parallel: videos shard across ranks (parallel.shard_videos), gradients through parallel.GradBuckets
(bucketed async all-reduce over RCCL), SyncBN statistics inside the selector.  One JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch (videos) for strong scaling, per-GPU for weak")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from anomalyclip_amd import init_weights as IW, parallel
    from anomalyclip_amd.anomaly_clip_module import AnomalyCLIPModule
    from anomalyclip_amd.components.anomaly_clip import AnomalyCLIP, lookup_prompts
    from anomalyclip_amd.components.loss import ComputeLoss
    from anomalyclip_amd.optim import AcxAdamW

    hc = IW.UCF_HEAD
    toks = torch.tensor(lookup_prompts(key="ucf")["tokenized_prompts"], dtype=torch.int32)
    net = AnomalyCLIP(arch="ViT-B/16", labels_key="ucf", emb_size=256, depth=1, heads=8, dim_heads=None, num_segments=32,
                      seg_length=16, concat_features=False, normal_id=7, stride=1, load_from_features=True,
                      select_idx_dropout_topk=0.7, select_idx_dropout_bottomk=0.7, ncrops=1, num_topk=3, num_bottomk=3)
    net.load_state_dict(IW.init_anomalyclip_state_dict(IW.VIT_B16, hc, toks, seed=0), strict=True)
    crit = ComputeLoss(7, 3, 1.0, 1.0, 1.0, 1.0, 1.0, 8e-4, 8e-3, 16, 32)
    mod = AnomalyCLIPModule(net, None, None, crit, num_classes=14, solver={"lr": 1e-5}).to(dev)
    mod.ncentroid = torch.zeros(512, device=dev)
    opt = mod.configure_optimizers()["optimizer"]

    B_global = args.batch if args.scaling == "strong" else args.batch * world
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B_global, 1, 512, 512, generator=g) * 0.3
    labels = torch.tensor([i % 13 + (1 if i % 13 >= 7 else 0) for i in range(B_global // 2)] + [7] * (B_global // 2))
    idx = parallel.shard_videos(B_global, world, rank)
    f_loc, l_loc = feats[idx].to(dev), labels[idx].to(dev)
    h = len(idx) // 2
    batch = ((f_loc[h:], l_loc[h:]), (f_loc[:h], l_loc[:h]))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    net.train()
    step_i = [0]

    def train_step():
        torch.manual_seed(step_i[0])          # host mask RNG, identical on every rank then sharded implicitly by video
        step_i[0] += 1
        mod.train_batch(batch, opt)

    def fwd_step():
        with torch.no_grad():
            net.eval()
            x = torch.cat((batch[1][0], batch[0][0]), 0)
            net(x.view(-1, 1, 512, 512), None, mod.ncentroid, 1, True)
            net.train()

    dt_train = timed(train_step)
    dt_fwd = timed(fwd_step)
    if rank == 0:
        feats_per_step = B_global * 512
        print(json.dumps({
            "metric": "features/sec through the head (UCF-Crime shape, 512-d), train step = fwd+loss+bwd+AdamW",
            "train_features_per_s": round(feats_per_step * args.steps / dt_train, 1),
            "train_ms_per_step": round(dt_train / args.steps * 1e3, 3),
            "fwd_features_per_s": round(feats_per_step * args.steps / dt_fwd, 1),
            "fwd_ms_per_step": round(dt_fwd / args.steps * 1e3, 3),
            "n_gpus": world, "global_batch_videos": B_global, "scaling": args.scaling, "dtype": "f32",
            "loss": float(mod.last_losses[0]), "data": "synthetic"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
