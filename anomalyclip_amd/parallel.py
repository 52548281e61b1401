"""Data-parallel glue: one process per GPU, torch.distributed ("nccl" == RCCL on
ROCm, "gloo" in the CPU tests).  The reference gets all of this implicitly from Lightning's
`strategy: ddp` + `sync_batchnorm: True` (configs/trainer/ddp.yaml:4,9); here the exchange steps are
explicit and sized for xGMI (point-to-point links, ~153 GB/s each):

  * GradBuckets      -- the 10.4 M trainable fp32 values (41.7 MB at the UCF config) live in ONE flat
                        buffer cut into a few large buckets in reverse-forward order; each bucket is
                        all-reduced asynchronously as soon as its last gradient has been produced
                        (overlaps with the remaining backward), averaged at finish().
  * sync_bn_stats    -- SyncBatchNorm statistics of the selector's BatchNorm1d(C-1): one all_gather of
                        (mean, M2, n) per rank, combined with Chan's parallel formula.
  * all_reduce_sum_  -- backward sums of the same BatchNorm.
  * shard_videos     -- contiguous, abnormal/normal-balanced split of a batch across ranks (the selector
                        assumes the first half of the local batch is abnormal, selector_model.py:132-156).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import contextlib
import os

import torch
import torch.distributed as dist


_LOCAL_ONLY = [False]


@contextlib.contextmanager
def local_only():
    """Inside an initialised process group, run this rank as if it were alone (no collective is issued, nothing is
    sharded): bench.py times T1 -- the single-GPU step -- in the same N-rank run that times TN."""
    keep = _LOCAL_ONLY[0]
    _LOCAL_ONLY[0] = True
    try:
        yield
    finally:
        _LOCAL_ONLY[0] = keep


def is_distributed() -> bool:
    if _LOCAL_ONLY[0]:
        return False
    # ACX_FORCE_COLLECTIVES=1 (tests): take the collective code paths in a 1-rank group too -- the only way a single-GPU box
    # can push the step's all-reduce / all-gather calls through the real RCCL backend
    return dist.is_available() and dist.is_initialized() and (
        dist.get_world_size() > 1 or os.environ.get("ACX_FORCE_COLLECTIVES") == "1")


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def shard_range(n: int, world: int, r: int):
    """contiguous block [lo, hi) of n items owned by rank r; the first n % world ranks get one more."""
    q, rem = divmod(n, world)
    lo = r * q + min(r, rem)
    return lo, lo + q + (1 if r < rem else 0)


def assemble_rows(local: torch.Tensor, lo: int, n: int) -> torch.Tensor:
    """every rank contributes rows [lo, lo+len(local)) of an [n, ...] tensor; returns the full tensor on every rank.
    One all-reduce of a zero-filled buffer (x + 0 == x exactly): at (14, 512) f32 = 28 KB the exchange is pure
    latency, and a single collective type works on RCCL and on gloo alike."""
    full = local.new_zeros((n,) + tuple(local.shape[1:]))
    full[lo:lo + local.shape[0]] = local
    return all_reduce_sum_(full)


def assemble_rows_into(full: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    """assemble_rows on a STATIC buffer (the step graphs): rows [lo, hi) of `full` hold this rank's block, every other row is
    zero; one in-place all-reduce leaves the complete tensor on every rank."""
    return all_reduce_sum_(full)


def all_gather_into(out: torch.Tensor, local: torch.Tensor) -> torch.Tensor:
    """out[r] = rank r's `local` (out: [world, n] static buffer); the list form works on RCCL and on gloo alike"""
    dist.all_gather(list(out.unbind(0)), local)
    return out


def combine_bn_stats(means: torch.Tensor, m2s: torch.Tensor, counts: torch.Tensor):
    """Chan et al. parallel variance: means/m2s [R, C], counts [R] -> (mean, var_biased, var_unbiased, n)."""
    n = counts.sum()
    w = (counts / n).unsqueeze(1)
    mean = (means * w).sum(0)
    m2 = (m2s + counts.unsqueeze(1) * (means - mean) ** 2).sum(0)
    return mean, m2 / n, m2 / (n - 1).clamp(min=1), n


def sync_bn_stats(mean: torch.Tensor, var_b: torch.Tensor, rows: int):
    """Local (mean, biased var, rows) -> global (mean, var_biased, var_unbiased, total_rows).  Nothing here synchronises the
    host with the device: the row count is written by a fill kernel (`new_tensor([..])` is a blocking pageable copy that
    waits for everything queued on the stream -- 2 ms in the middle of every forward) and the TOTAL stays a device scalar
    (ops.bn_bwd_apply reads it there) unless the tensors live on the CPU."""
    C1 = mean.numel()
    local = torch.cat([mean, var_b * rows, torch.full((1,), float(rows), dtype=mean.dtype, device=mean.device)])
    gathered = [torch.empty_like(local) for _ in range(world_size())]
    dist.all_gather(gathered, local)
    g = torch.stack(gathered)
    if g.is_cuda:                                   # one libacx launch instead of a dozen elementwise ones
        from . import ops
        return ops.bn_combine(g, C1)
    m, vb, vu, n = combine_bn_stats(g[:, :C1], g[:, C1:2 * C1], g[:, 2 * C1])
    total = n.reshape(1).contiguous() if n.is_cuda else int(round(float(n)))
    return m.contiguous(), vb.contiguous(), vu.contiguous(), total


def shard_videos(batch: int, world: int, r: int) -> List[int]:
    """Indices of the videos of a [abnormal..., normal...] batch that rank r processes, again ordered
    [abnormal..., normal...]; every rank gets batch/(2*world) of each."""
    half = batch // 2
    if half % world:
        raise ValueError(f"batch/2 = {half} videos per class is not divisible by world size {world}")
    per = half // world
    a = list(range(r * per, (r + 1) * per))
    n = list(range(half + r * per, half + (r + 1) * per))
    return a + n


class GradBuckets:
    """Flat gradient buffer + bucketed asynchronous all-reduce.

    params: trainable parameters in FORWARD order; buckets are filled in reverse order (the order autograd
    produces gradients) and launched as soon as complete.  Every param.grad is a view into the flat buffer."""

    def __init__(self, params: Sequence[torch.nn.Parameter], bucket_bytes: int = 16 << 20, cuts: Sequence[torch.nn.Parameter] = ()):
        """cuts: parameters at which a bucket must END (walking the list from the back, i.e. in the order gradients arrive):
        [ctx, text_projection] get a bucket of their own, so the step graph can exchange the text gradients on the text
        stream as soon as the text backward is done, independently of the temporal model's buckets."""
        self.params = [p for p in params if p.requires_grad]
        cut_ids = {id(p) for p in cuts}
        # every view starts on a 256-byte boundary: kernels that PRODUCE a gradient in its view (acx_gemm_tn's `C`, the step
        # graph) need 16-byte alignment, and full-line starts cost 63 pad floats per tensor at most
        ALIGN = 64
        total = sum((p.numel() + ALIGN - 1) // ALIGN * ALIGN for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets: List[List[int]] = []          # param indices per bucket
        self.ranges: List[tuple] = []
        # assign offsets in reverse-forward order so each bucket is a contiguous slice
        off = 0
        cur: List[int] = []
        cur_start = 0
        self._views = {}
        for i in reversed(range(len(self.params))):
            p = self.params[i]
            self._views[i] = (off, off + p.numel())
            p.grad = self.view(i)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            cur.append(i)
            nxt = self.params[i - 1] if i > 0 else None
            if (off - cur_start) * 4 >= bucket_bytes or (nxt is not None and id(nxt) in cut_ids):
                self.buckets.append(cur)
                self.ranges.append((cur_start, off))
                cur, cur_start = [], off
        if cur:
            self.buckets.append(cur)
            self.ranges.append((cur_start, off))
        self._bucket_of = {i: b for b, idxs in enumerate(self.buckets) for i in idxs}
        self._pending = [len(b) for b in self.buckets]
        self._handles = []
        self._fired = set()
        self._armed = False          # between zero() and finish(): the only window in which hooks / accumulate() act
        self._hook_fns = [self._make_hook(i) for i in range(len(self.params))]
        self._hooks = [p.register_post_accumulate_grad_hook(fn) for p, fn in zip(self.params, self._hook_fns)]

    def view(self, i: int) -> torch.Tensor:
        """Parameter i's slice of the flat buffer with the PARAMETER's strides: a dense parameter that is not row-major (the
        channels-last conv weights of the temporal model) gets a gradient with the same physical element order, so AdamW's
        elementwise pass over the raw memory of (p, g, m, v) lines up and a kernel can produce the gradient in place."""
        p = self.params[i]
        lo, hi = self._views[i]
        if p.is_contiguous():
            return self.flat[lo:hi].view_as(p)
        if not _is_dense(p):
            raise ValueError("GradBuckets: parameters must be dense (contiguous in some dimension order)")
        return torch.as_strided(self.flat, p.shape, p.stride(), lo)

    def _make_hook(self, i):
        def hook(p):
            if not self._armed:
                # a backward outside zero() ... finish() (Lightning's automatic optimisation on the same module, a test calling
                # training_step + backward): plain autograd semantics, no bucket bookkeeping, no collective
                return
            if i in self._fired:
                # second report of the same parameter in one step: accumulate() reported it and autograd calls the
                # post-accumulate hook of its AccumulateGrad node anyway (with an undefined gradient: nothing was added)
                return
            lo, hi = self._views[i]
            if p.grad is not None and p.grad.data_ptr() != self.flat[lo:hi].data_ptr():
                # autograd replaced the view (first accumulation into a None grad): copy back and re-point
                v = self.view(i)
                v.copy_(p.grad)
                p.grad = v
            self._fired.add(i)
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and is_distributed():
                lo_b, hi_b = self.ranges[b]
                self._handles.append(dist.all_reduce(self.flat[lo_b:hi_b], op=dist.ReduceOp.SUM, async_op=True))
        return hook

    # ---- gradient sink: a backward function that already holds ALL gradients of a group of parameters (the graph-replayed
    # temporal model) adds them into the flat buffer with ONE launch and reports them ready, instead of returning them to
    # autograd (one AccumulateGrad launch + one hook per parameter)
    def index_of(self, p):
        if not hasattr(self, "_index"):
            self._index = {id(q): i for i, q in enumerate(self.params)}
        return self._index.get(id(p))

    def accumulate(self, params, grads) -> bool:
        """params[k].grad += grads[k] for every pair with a gradient, bucket bookkeeping included.  Returns False (and does
        nothing) unless every such parameter is one of this buffer's and still points at its view."""
        if not self._armed:
            return False
        pairs = [(p, g) for p, g in zip(params, grads) if g is not None]
        idx = [self.index_of(p) for p, _ in pairs]
        if any(i is None for i in idx):
            return False
        views, srcs = [], []
        for (p, g), i in zip(pairs, idx):
            lo, hi = self._views[i]
            # same PHYSICAL element order on both sides: the gradient must carry the parameter's (dense) strides
            if p.grad is None or p.grad.data_ptr() != self.flat[lo:hi].data_ptr() or not same_layout(g, p.grad):
                return False
            views.append(self.flat[lo:hi])
            srcs.append(torch.as_strided(g, (g.numel(),), (1,), g.storage_offset()))
        from . import ops
        ops.multi_axpy_(views, srcs, 1.0)
        hooks = self._hook_fns
        for (p, _), i in zip(pairs, idx):
            hooks[i](p)
        return True

    def zero(self):
        self.flat.zero_()
        self.arm()

    def arm(self):
        """start of a step WITHOUT zeroing the flat buffer (the whole-step graph overwrites every gradient it produces)"""
        self._pending = [len(b) for b in self.buckets]
        self._handles = []
        self._fired = set()
        self._armed = True
        for i, p in enumerate(self.params):          # finish() detaches the grads of unused parameters
            if p.grad is None:
                p.grad = self.view(i)

    def reduce_now(self, params) -> None:
        """the gradients of `params` -- together exactly one or more WHOLE buckets -- are complete: all-reduce those buckets on
        the CURRENT stream (blocking collective, stream-ordered) and settle their bookkeeping, so that neither the hooks nor
        finish() touch them again"""
        bs = set()
        for p in params:
            i = self.index_of(p)
            if i is None:
                raise ValueError("reduce_now: not a parameter of this buffer")
            self._fired.add(i)
            bs.add(self._bucket_of[i])
        for b in sorted(bs):
            if not set(self.buckets[b]) <= {self.index_of(p) for p in params}:
                raise ValueError("reduce_now: the parameters do not cover their bucket")
            self._pending[b] = 0
            if is_distributed():
                lo, hi = self.ranges[b]
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)

    def mark_ready(self, params):
        """the step graph produced these parameters' gradients in their views: bucket bookkeeping (and, under data
        parallelism, the all-reduce of every bucket that became complete)"""
        for p in params:
            i = self.index_of(p)
            if i is not None:
                self._hook_fns[i](p)

    def finish(self, average: bool = True):
        """Wait for the in-flight all-reduces, reduce buckets whose gradients never arrived (unused
        parameters, e.g. selector_model.logit_scale), and average (average=False leaves the SUM: the step graph's AdamW
        multiplies by 1 / world as it reads the gradients)."""
        # A parameter that received no gradient keeps grad = None, as under the reference's DDP
        # (find_unused_parameters: a globally unused parameter's .grad is left untouched), so AdamW skips it --
        # no weight decay on the never-used selector_model.logit_scale (selector_model.py:22).  The autograd graph
        # is the same on every rank, hence so is the set of unused parameters.
        self._armed = False
        for i, p in enumerate(self.params):
            if i not in self._fired:
                p.grad = None
        if not is_distributed():
            return
        for b, pend in enumerate(self._pending):
            if pend > 0:
                lo, hi = self.ranges[b]
                self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        for h in self._handles:
            h.wait()
        if average:
            self.flat.div_(world_size())
        self._handles = []


def _is_dense(t: torch.Tensor) -> bool:
    """non-overlapping and dense: some permutation of the dimensions is contiguous"""
    dims = sorted(range(t.dim()), key=lambda d: (t.stride(d), t.shape[d]), reverse=True)
    return t.permute(*dims).is_contiguous() if t.dim() else True


def same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """same shape and the same physical element order (strides compared on the dimensions that have more than one element)"""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


def rank_zero_only(fn):
    """pytorch_lightning.utilities.rank_zero_only as the reference applies it to test_step / test_epoch_end
    (anomaly_clip_module.py:458,500): the body runs on rank 0, other ranks return None."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if rank() == 0:
            return fn(*args, **kwargs)
        return None
    return wrapped
