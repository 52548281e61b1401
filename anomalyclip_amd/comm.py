"""libacx's own collectives (include/acx.h: acx_comm_init / acx_allreduce / acx_allgather): thin RCCL calls on the caller's
stream, the torch-free form of the exchange the reference runs through Lightning DDP over NCCL (configs/trainer/ddp.yaml:1-9).

The package's training path keeps torch.distributed (backend "nccl" = RCCL) as its default transport -- anomalyclip_amd/parallel.py --
because that is what the reference's Trainer hands a module and what the driver launches (`python -m torch.distributed.run`).  This
module is the alternative for hosts without torch.distributed and for HIP graphs: a collective issued here on a capturing stream is
recorded into the graph like a kernel (torch's ProcessGroup hops to its own stream).  The 128-byte RCCL id travels over whatever the
host has -- here: an object broadcast of an already initialised torch.distributed group of ANY backend (gloo is enough), or a file.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from . import ops

_DT = {torch.float32: L.ACX_F32, torch.bfloat16: L.ACX_BF16, torch.float64: L.ACX_F64, torch.int64: L.ACX_I64}
_OPS = {"sum": L.COMM_SUM, "max": L.COMM_MAX, "min": L.COMM_MIN}


def unique_id() -> bytes:
    """rank 0: a fresh RCCL id (ncclGetUniqueId) to hand to every rank"""
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    L.check(L.lib().acx_comm_unique_id(buf, L.COMM_ID_BYTES), None)
    return buf.raw


def init(device_index: int, rank: int, world: int, uid: Optional[bytes] = None) -> None:
    """Collective over all ranks: creates this device context's communicator.  uid = None: rank 0 draws the id and it is broadcast
    through the initialised torch.distributed group (any backend)."""
    if uid is None:
        if world == 1:
            uid = unique_id()
        else:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise L.AcxError("comm.init: pass the 128-byte id (comm.unique_id() of rank 0), or initialise torch.distributed to carry it")
            box = [unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
    assert len(uid) == L.COMM_ID_BYTES
    h = L.ctx(device_index)
    L.check(L.lib().acx_comm_init(h, int(rank), int(world), C.c_char_p(uid)), h)


def destroy(device_index: int) -> None:
    h = L.ctx(device_index)
    L.check(L.lib().acx_comm_destroy(h), h)


def info(device_index: int):
    """(rank, world) of the context's communicator; world = 0: none"""
    h = L.ctx(device_index)
    r, w = C.c_int32(0), C.c_int32(0)
    L.check(L.lib().acx_comm_info(h, C.byref(r), C.byref(w)), h)
    return int(r.value), int(w.value)


def all_reduce(t: torch.Tensor, op: str = "sum") -> torch.Tensor:
    """in place on the current stream (ordered with the library's kernels; capturable)"""
    assert t.is_cuda and t.is_contiguous() and t.dtype in _DT
    h = ops._h(t)
    L.check(L.lib().acx_allreduce(h, t.data_ptr(), t.numel(), _DT[t.dtype], _OPS[op], ops._stream()), h)
    return t


def all_gather(out: torch.Tensor, local: torch.Tensor) -> torch.Tensor:
    """out[world * n] <- every rank's local[n], rank-major, on the current stream"""
    assert out.is_cuda and local.is_cuda and out.is_contiguous() and local.is_contiguous() and out.dtype == local.dtype in _DT
    assert out.numel() % max(1, local.numel()) == 0
    h = ops._h(local)
    L.check(L.lib().acx_allgather(h, local.data_ptr(), out.data_ptr(), local.numel(), _DT[local.dtype], ops._stream()), h)
    return out
