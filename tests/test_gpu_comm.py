"""-m gpu: libacx's own collectives (include/acx.h: acx_comm_init / acx_allreduce / acx_allgather -- thin RCCL calls on the caller's
stream).  A single-GPU box runs them through RCCL itself in a 1-rank communicator (eagerly and recorded into
a HIP graph); with two GPUs visible the same calls run with two ranks and are compared with hand-computed sums."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from anomalyclip_amd import _lib as L
from anomalyclip_amd import comm

DEV = "cuda"


def test_comm_single_rank_eager_and_captured():
    di = torch.cuda.current_device()
    assert comm.info(di) == (0, 0)
    with pytest.raises(L.AcxError):
        comm.all_reduce(torch.ones(4, device=DEV))                    # no communicator yet: an error, never a silent no-op
    comm.init(di, 0, 1)
    try:
        assert comm.info(di) == (0, 1)
        with pytest.raises(L.AcxError):
            comm.init(di, 0, 1)                                        # one communicator per context
        for dt in (torch.float32, torch.bfloat16, torch.float64, torch.int64):
            x = (torch.arange(1000, device=DEV) % 7).to(dt)
            y = x.clone()
            comm.all_reduce(y)
            comm.all_reduce(y, "max")
            assert torch.equal(x, y), dt
        loc = torch.randn(333, device=DEV)
        out = torch.zeros(333, device=DEV)
        comm.all_gather(out, loc)
        assert torch.equal(out, loc)
        # the collective orders with the library's kernels on the caller's stream, and a capturing stream records it: the
        # gradient all-reduce can sit INSIDE a replayed step graph (torch's ProcessGroup hops to a stream of its own)
        from anomalyclip_amd import ops
        a = torch.randn(256, 512, device=DEV)
        w = torch.randn(128, 512, device=DEV)
        y = torch.empty(256, 128, device=DEV)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.gemm(a, w, out=y)
            comm.all_reduce(y)
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                ops.gemm(a, w, out=y)
                comm.all_reduce(y)
        torch.cuda.current_stream().wait_stream(s)
        want = y.clone()
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want) and float(want.abs().max()) > 0
    finally:
        comm.destroy(di)
    assert comm.info(di) == (0, 0)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)   # carries the 128-byte id only
    ok = True
    try:
        comm.init(rank, rank, world)
        x = torch.full((4096,), float(rank + 1), device="cuda")
        comm.all_reduce(x)
        ok &= bool((x == float(world * (world + 1) // 2)).all())
        m = torch.tensor([rank, 10 - rank], device="cuda", dtype=torch.int64)
        comm.all_reduce(m, "max")
        ok &= m.tolist() == [world - 1, 10]
        loc = torch.full((5,), float(rank), device="cuda")
        out = torch.empty(5 * world, device="cuda")
        comm.all_gather(out, loc)
        ok &= out.view(world, 5)[:, 0].tolist() == [float(r) for r in range(world)]
        torch.cuda.synchronize()
        comm.destroy(rank)
    finally:
        q.put((rank, bool(ok)))
        dist.barrier()
        dist.destroy_process_group()


def test_comm_two_ranks_when_two_gpus_are_visible():
    """self-arming: runs the day the box shows two GPUs (the driver's multi-GPU node); skips on the single-GPU boxes"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL)")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)], res
