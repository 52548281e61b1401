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
