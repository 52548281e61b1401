"""CPU, gloo, world_size 2: the data-parallel exchange steps (bucketed gradient all-reduce, SyncBN statistics
all-gather, ncentroid reduction) without any GPU kernel."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anomalyclip_amd import parallel
    torch.manual_seed(0)
    # --- GradBuckets: 5 parameters, tiny bucket size -> several buckets; one parameter never gets a gradient
    params = [torch.nn.Parameter(torch.randn(n)) for n in (10, 300, 7, 50, 1)]
    gb = parallel.GradBuckets(params, bucket_bytes=256)
    gb.zero()
    loss = sum((p * (rank + 1) * (i + 1)).sum() for i, p in enumerate(params[:4]))     # params[4] unused
    loss.backward()
    gb.finish()
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params[:4]))
    ok &= params[4].grad is None                   # unused parameter: grad stays None (the reference's DDP semantics)
    ok &= len(gb.buckets) >= 2
    # second step reuses the flat buffer
    gb.zero()
    sum((p * (rank + 3)).sum() for p in params[:4]).backward()
    gb.finish()
    ok &= all(torch.allclose(p.grad, torch.full_like(p, 3.5)) for p in params[:4])
    ok &= all(p.grad.data_ptr() == gb.flat[gb._views[i][0]:].data_ptr() for i, p in enumerate(params[:4]))   # still views
    ok &= params[4].grad is None
    # --- SyncBN statistics: each rank holds a different slab of rows
    g = torch.Generator().manual_seed(1)
    x = torch.randn(700, 13, generator=g) * 2 + 0.5
    mine = x[:300] if rank == 0 else x[300:]
    m, vb, vu, n = parallel.sync_bn_stats(mine.mean(0), mine.var(0, unbiased=False), mine.shape[0])
    ok &= torch.allclose(m, x.mean(0), atol=1e-5) and torch.allclose(vb, x.var(0, unbiased=False), atol=1e-5)
    ok &= torch.allclose(vu, x.var(0, unbiased=True), atol=1e-5) and n == 700
    # --- ncentroid-style (sum, count) reduction
    s = mine.sum(0)
    parallel.all_reduce_sum_(s)
    ok &= torch.allclose(s, x.sum(0), atol=1e-3)
    # --- sharding covers the batch exactly once
    idx = parallel.shard_videos(8, world, rank)
    ok &= idx == ([0, 1, 4, 5] if rank == 0 else [2, 3, 6, 7])
    # --- class-parallel text features (functional.TextFeaturesFn): the exchange logic with the libacx row
    # functions swapped for a torch toy of the same contract (rows independent per class).  Expected result: the
    # data-parallel mean over ranks of the gradients of each rank's own loss, as the reference's DDP produces.
    ok &= [parallel.shard_range(14, 8, r) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12), (12, 13), (13, 14)]
    from types import SimpleNamespace
    from anomalyclip_amd.components import functional as Fn
    C, d, e = 5, 6, 4
    gen = torch.Generator().manual_seed(7)
    ctx0, P0 = torch.randn(C, 3, d, generator=gen), torch.randn(d, e, generator=gen)
    wr = [torch.randn(C, e, generator=gen) for _ in range(world)]           # rank r's loss = sum(T * wr[r])

    def toy_fwd(net, ctx_param, P, lo, hi):
        return torch.tanh(ctx_param.detach()[lo:hi].sum(1)) @ P.detach(), (lo, hi)

    def toy_bwd(net, P, state, d_tf):
        lo, hi = state
        with torch.enable_grad():
            c = net.prompt_learner.ctx.detach()[lo:hi].clone().requires_grad_(True)
            Pp = P.detach().clone().requires_grad_(True)
            (torch.tanh(c.sum(1)) @ Pp).backward(d_tf)
        return c.grad, Pp.grad

    Fn._text_forward_rows, Fn._text_backward_rows = toy_fwd, toy_bwd
    ctxp, Pp = torch.nn.Parameter(ctx0.clone()), torch.nn.Parameter(P0.clone())
    net = SimpleNamespace(prompt_learner=SimpleNamespace(n_cls=C, ctx=ctxp))
    gb2 = parallel.GradBuckets([ctxp, Pp])
    gb2.zero()
    T = Fn.TextFeaturesFn.apply(ctxp, Pp, net, True)
    (T * wr[rank]).sum().backward()
    gb2.finish()
    c_ref, P_ref = ctx0.clone().requires_grad_(True), P0.clone().requires_grad_(True)
    T_ref = torch.tanh(c_ref.sum(1)) @ P_ref
    (sum((T_ref * w).sum() for w in wr) / world).backward()
    ok &= torch.allclose(T.detach(), T_ref.detach(), atol=1e-6)
    ok &= torch.allclose(ctxp.grad, c_ref.grad, atol=1e-6) and torch.allclose(Pp.grad, P_ref.grad, atol=1e-6)
    # --- more ranks than classes (XD-Violence: 7 classes on 8 GPUs): the rank that owns no class contributes an empty
    # block and zero gradients but joins both exchanges (a raise or a skipped collective would hang the other rank)
    assert parallel.shard_range(1, world, 1) == (1, 1)
    ctx1, w1 = ctx0[:1].clone(), [w[:1] for w in wr]
    ctxq, Pq = torch.nn.Parameter(ctx1.clone()), torch.nn.Parameter(P0.clone())
    net1 = SimpleNamespace(prompt_learner=SimpleNamespace(n_cls=1, ctx=ctxq))
    gb3 = parallel.GradBuckets([ctxq, Pq])
    gb3.zero()
    T1 = Fn.TextFeaturesFn.apply(ctxq, Pq, net1, True)
    (T1 * w1[rank]).sum().backward()
    gb3.finish()
    c_ref, P_ref = ctx1.clone().requires_grad_(True), P0.clone().requires_grad_(True)
    T_ref = torch.tanh(c_ref.sum(1)) @ P_ref
    (sum((T_ref * w).sum() for w in w1) / world).backward()
    ok &= T1.shape == (1, e) and torch.allclose(T1.detach(), T_ref.detach(), atol=1e-6)
    ok &= torch.allclose(ctxq.grad, c_ref.grad, atol=1e-6) and torch.allclose(Pq.grad, P_ref.grad, atol=1e-6)
    # --- the built-in Trainer shards torch DataLoaders like Lightning's DDP strategy (DistributedSampler + set_epoch)
    from torch.utils.data import DataLoader, TensorDataset
    from anomalyclip_amd.trainer import Trainer
    dl = DataLoader(TensorDataset(torch.arange(10)), batch_size=1, shuffle=True)
    seen = [sorted(int(b[0]) for b in Trainer._shard_loader(dl, ep)) for ep in (0, 1)]
    allv = [torch.zeros(10), torch.zeros(10)]
    for ep in (0, 1):
        allv[ep][seen[ep]] = 1
        dist.all_reduce(allv[ep])
    ok &= len(seen[0]) == 5 and bool((allv[0] == 1).all()) and bool((allv[1] == 1).all())     # disjoint cover, every epoch
    tb = list(Trainer()._train_batches([dl, DataLoader(TensorDataset(torch.arange(4)), batch_size=1)], 0))
    ok &= len(tb) == 5                              # max_size_cycle over the SHARDED lengths (5 and 2)
    # --- a meter that saw no update on this rank still joins the epoch-end exchange
    from anomalyclip_amd.anomaly_clip_module import MeanMetric
    mm = MeanMetric()
    if rank == 0:
        mm.update(torch.tensor(3.0))
    ok &= abs(float(mm.compute()) - 3.0) < 1e-6
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)], res
