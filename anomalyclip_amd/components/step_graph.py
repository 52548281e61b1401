"""Whole-step training graph: one optimisation step of `AnomalyCLIPModule.train_batch` -- forward, the 7-term loss, backward,
gradient exchange and AdamW (what Lightning's automatic optimisation + DDP do around `training_step`,
anomaly_clip_module.py:203-293,693-746; configs/trainer/ddp.yaml:1-9) -- as a FIXED sequence of libacx launches on static
buffers, captured once into HIP graphs and replayed every step.

Why: with few videos per GPU (a data-parallel rank's share of a 64-video batch) the step is bound by launch latency and by
the host, not by arithmetic.  The autograd path issues ~700 library calls and ~80 torch glue kernels (cat / fill / copy /
clone / accumulate) per step from Python; here the host issues a handful of graph replays and the device sees

  * no torch kernel between the first and the last launch of a step (inputs arrive by three copies into static buffers);
  * every gradient produced IN its view of the flat gradient buffer (weight-gradient GEMMs write the view; the small rest
    by one multi-copy launch), no zero-fill, no per-parameter accumulate;
  * two LINEAR launch chains on two streams, no fork inside any graph: the text tower (forward graph, backward graph) on
    its own stream beside the temporal model + selector + loss chain on the caller's stream; the selector's short
    backward runs before the temporal backward so that the text backward overlaps the latter;
  * AdamW inside the graph with its per-tensor scalars in device memory (lr schedules do not re-capture).

The launch sequence calls the same forward / backward functions as the autograd path (functional.TemporalFn,
_text_forward_rows / _text_backward_rows, the selector and loss kernels) on the same operands in the same per-stream order,
so losses, gradients and updated weights are bit-identical to it (tests/test_gpu_train.py).  Under data parallelism the
sequence is cut into graph segments at the exchange steps (text features, SyncBN statistics forward / backward sums, text
feature gradients, the gradient buffer), which run eagerly on static buffers in between.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, List, Optional

import torch

from .. import _lib as L
from .. import ops
from . import functional as Fn


def _par():
    from .. import parallel
    return parallel


import os as _os
# development probe (timing only, results are garbage): replay the main chain without the text stream's graphs, to see what the
# text tower still costs the step when it runs beside it
_SKIP_TEXT = _os.environ.get("ACX_STEP_SKIP_TEXT") == "1"
# development probe: host time stamps of every replayed item of every step (ACX_STEP_HOST_TRACE=1; read step_graph._HOST_TRACE)
_HOST_TRACE = [] if _os.environ.get("ACX_STEP_HOST_TRACE") == "1" else None
# development A/B: the step's middle section as the separate launches of the autograd path (no acx_selector_tail / acx_mil_loss_bn)
_UNFUSED = _os.environ.get("ACX_STEP_UNFUSED") == "1"


class _Program:
    """Records the step as a list of items while the sequence function runs: graph segments on the main stream, eager
    callables (collectives) between them, and graph segments on the temporal-backward stream.  `mode`:
      "eager"   -- warm-up: everything runs directly, nothing is recorded (lazy workspaces, kernel attributes, tables);
      "capture" -- segments are captured; eager items run once (the capture-time data flow stays real) and are recorded."""

    def __init__(self, mode: str, cap: torch.cuda.Stream, tstream: torch.cuda.Stream):
        # `cap`: the stream the main segments are CAPTURED on (a capture cannot run on the default stream); they replay on
        # whatever stream is current in step().  `tstream`: capture AND replay stream of the temporal backward.
        self.mode, self.cap, self.tstream = mode, cap, tstream
        self.items: List[tuple] = []
        self._g = None
        self._cm = None
        self.pool = None
        self.xpool = None
        self.n_marks = 0
        self.keep: List[object] = []           # tensors that must outlive the capture (read across segments / streams)

    # ---- main-stream segments
    def begin(self):
        """(re)open a main-stream segment; idempotent.  eager() / branch() / join() close the open segment and leave it to the
        sequence to call begin() before its next launch (no empty graphs between two adjacent cuts)."""
        if self.mode != "capture" or self._cm is not None:
            return
        self._g = torch.cuda.CUDAGraph()
        kw = {"pool": self.pool} if self.pool is not None else {}
        self._cm = torch.cuda.graph(self._g, stream=self.cap, capture_error_mode="thread_local", **kw)
        self._cm.__enter__()

    def end(self):
        if self.mode != "capture" or self._cm is None:
            return
        self._cm.__exit__(None, None, None)
        if self.pool is None:
            self.pool = self._g.pool()
        self.items.append(("graph", self._g, "main"))
        self._g = self._cm = None

    def abort(self):
        if self._cm is not None:
            try:
                self._cm.__exit__(RuntimeError, RuntimeError("capture aborted"), None)
            except Exception:  # noqa: BLE001
                pass
            self._g = self._cm = None

    def eager(self, fn: Callable[[], None]):
        """an exchange step between two graph segments (operates on static buffers only)"""
        if self.mode != "capture":
            fn()
            return
        self.end()
        fn()
        self.items.append(("eager", fn, "main"))

    # ---- the text tower: its own linear graphs on its own stream with their own memory pool (they replay concurrently with
    # main-stream segments, so they must not share the main pool).  No graph on either stream contains a fork: ROCm spreads
    # the parallel branches of ONE graph over several hardware queues, where they collide with the other stream's work
    # (measured: a forked temporal backward serialised the main stream's next segment behind it, profiles/r04_step_graph_notes.md)
    def mark(self):
        """a point in the main stream's order (closes the open segment); on_x(..., after=mark) makes a text-stream graph wait
        for the main-stream work BEFORE that point only -- the main segments recorded after it are enqueued first and run
        beside it"""
        if self.mode != "capture":
            return None
        self.end()
        k = self.n_marks
        self.n_marks += 1
        self.items.append(("record", k))
        return k

    def on_x(self, fn: Callable[[], None], after=None):
        """fn() as one graph on the text stream.  ENQUEUE ORDER MATTERS on ROCm: hipGraphLaunch walks the graph on the host
        (~3.5 us per node), so the 170-node text graphs are replayed AFTER the short-to-enqueue, long-running main segment
        they overlap with (recorded between mark() and this call) -- launched first they left the GPU with nothing but
        5-us text kernels for 0.6 ms of every step."""
        if self.mode != "capture":
            fn()
            return
        self.end()
        if after is None:
            after = self.mark()
        self.items.append(("x_wait", after))
        self.tstream.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        kw = {"pool": self.xpool} if self.xpool is not None else {}
        with torch.cuda.graph(g, stream=self.tstream, capture_error_mode="thread_local", **kw):
            fn()
        if self.xpool is None:
            self.xpool = g.pool()
        self.items.append(("graph", g, "x"))

    def x_eager(self, fn: Callable[[], None]):
        """an exchange step on the TEXT stream (between two of its graphs)"""
        if self.mode != "capture":
            fn()
            return
        with torch.cuda.stream(self.tstream):
            fn()
        self.items.append(("x_eager", fn))

    def x_mark(self, name: str):
        """a named point in the text stream's order (main_wait(name) makes the main stream wait for it)"""
        if self.mode == "capture":
            self.items.append(("record_x", name))

    def main_wait(self, name: str):
        if self.mode != "capture":
            return
        self.end()
        torch.cuda.current_stream().wait_stream(self.tstream)
        self.items.append(("main_wait_x", name))

    def host(self, fn: Callable[[], None]):
        """pure host bookkeeping (no device work): runs now and at every replay, does not cut the graph segment"""
        fn()
        if self.mode == "capture":
            self.items.append(("host", fn))


class TrainStepGraph:
    """See the module docstring.  Built for one (module, optimizer, batch geometry); `step(batch)` runs one optimisation
    step and returns the 8 loss terms (views of a static tensor: read them before the next step)."""

    def __init__(self, module, optimizer, n_abnormal: int, n_normal: int, frames: int):
        from ..optim import AcxAdamW
        net = module.net
        self.mod, self.net, self.opt = module, net, optimizer
        self.dev = module.device
        par = _par()
        if not net.load_from_features or net.ncrops != 1 or net.precision not in ("f32", "auto"):
            raise L.AcxError("the step graph covers training from pre-extracted features, one crop, f32")
        if not (net.prompt_learner.ctx.requires_grad and net.text_encoder.text_projection.requires_grad):
            raise L.AcxError("the step graph expects trainable prompt context and text projection")
        if not all(p.requires_grad for p in net.temporal_model.parameters()):
            raise L.AcxError("the step graph expects every temporal-model parameter to be trainable")
        self.in_graph_adamw = isinstance(optimizer, AcxAdamW)
        sel, tm = net.selector_model, net.temporal_model
        self.B, self.ha, self.hn = n_abnormal + n_normal, n_abnormal, n_normal
        if self.ha != self.hn:
            raise L.AcxError("the selector assumes half of the batch abnormal, half normal (selector_model.py:132-156)")
        N, Lg = sel.num_segments, sel.seg_length
        if frames != N * Lg:
            raise L.AcxError("training batches hold num_segments * seg_length features per video")
        D = net.embedding_dim
        dev = self.dev
        self.x = torch.empty(self.B * N * Lg, D, dtype=torch.float32, device=dev)
        self.labels = torch.empty(self.B, dtype=torch.int64, device=dev)
        self.masks = torch.empty(2, self.B, N, dtype=torch.float32, device=dev)
        self._ring = 0
        self.masks_host = [torch.empty(2, self.B, N, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._ring_ev = [None] * 4
        self.buckets = module._buckets
        self.gviews = {p: self.buckets.view(i) for i, p in enumerate(self.buckets.params)}
        self.meter_sum = torch.zeros(8, dtype=torch.float32, device=dev)
        # ---- optimizer: tensors with a gradient in this step = everything trainable except the never-used logit_scale
        self.text_params = [net.prompt_learner.ctx, net.text_encoder.text_projection]
        self.main_params = [p for p in self.buckets.params if p is not sel.logit_scale and all(p is not q for q in self.text_params)]
        self.opt_params = self.text_params + self.main_params
        # hyper layout: [bc2 | text pairs | bc2 | temporal pairs]: two launches (text stream / main stream), one upload
        self.n_hyper = 2 + 2 * len(self.opt_params)
        self.hyper_host = [torch.empty(self.n_hyper, dtype=torch.float32).pin_memory() for _ in range(4)]
        self.hyper_dev = torch.zeros(self.n_hyper, dtype=torch.float32, device=dev)
        self.world = par.world_size()
        self.distributed = par.is_distributed()
        self.cap = torch.cuda.Stream(device=dev)          # capture stream of the main segments
        # the text tower's stream (capture and replay), high priority: its 5-us launches go first whenever a slot is free
        # -- on ANOTHER hardware queue than the stream the step replays on (ops.side_stream_beside: with a few more streams alive in
        # the process -- RCCL's, a loader's -- a fresh stream may share the main stream's queue, and the step's two chains serialise)
        self.tstream = (torch.cuda.Stream(device=dev, priority=-1) if _os.environ.get("ACX_STEP_PLAIN_TSTREAM") == "1"
                        else ops.side_stream_beside(torch.cuda.current_stream(), dev, priority=-1))
        self.ev_marks = [torch.cuda.Event() for _ in range(12)]
        self.ev_x = {"text_fwd": torch.cuda.Event(), "text_params": torch.cuda.Event()}
        self._text_state = None
        self.key = self.make_key(module, optimizer, n_abnormal, n_normal, frames)
        self.program = None

    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def make_key(module, optimizer, n_abnormal, n_normal, frames):
        net = module.net
        par = _par()
        nc = module.ncentroid
        x6 = ops.x6_options(torch.cuda.current_device())       # grid / K split of the captured bf16 x 6 launches
        return (n_abnormal, n_normal, frames, id(optimizer), par.is_distributed(), par.world_size(), par.rank(),
                int(x6["cus"]), bool(x6["tail_split"]),
                None if nc is None else nc.data_ptr(), bool(net.concat_features), int(getattr(net, "text_len", 0)),
                bool(getattr(net, "text_class_parallel", True)), torch.cuda.current_stream().cuda_stream,
                net.eot_index.data_ptr(), net.eot_index._version) + tuple(      # the EOT row table is baked into the capture
            p.data_ptr() for p in module._buckets.params)

    # ------------------------------------------------------------------------------------------------------------------
    def _opt_state(self):
        """AdamW state of every stepped parameter (created like AcxAdamW.step creates it), and (lr, wd) per parameter"""
        grp = {}
        for g in self.opt.param_groups:
            for p in g["params"]:
                grp[p] = g
        ms, vs, lrs, wds, betas, eps = [], [], [], [], None, None
        for p in self.opt_params:
            if p not in grp:
                raise L.AcxError("the optimizer does not hold every trainable parameter of the step")
            g = grp[p]
            st = self.opt.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            ms.append(st["exp_avg"])
            vs.append(st["exp_avg_sq"])
            lrs.append(float(g["lr"]))
            wds.append(float(g["weight_decay"]))
            b = (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
            if betas is not None and b != (betas[0], betas[1], eps):
                raise L.AcxError("the in-graph AdamW needs one (betas, eps) for all param groups")
            betas, eps = (b[0], b[1]), b[2]
        return ms, vs, lrs, wds, betas, eps

    def _refresh_hyper(self):
        """host side of the optimizer step: step counters, this step's (lr, weight decay) per tensor -> device scalars"""
        par = _par()
        ms, vs, lrs, wds, betas, eps = self._opt_state()
        steps = set()
        for p in self.opt_params:
            st = self.opt.state[p]
            st["step"] += 1
            steps.add(int(st["step"]))
        if len(steps) != 1:
            raise L.AcxError("the in-graph AdamW needs one step count for all parameters")
        k = self._slot()
        step, nt = steps.pop(), len(self.text_params)
        ops.adamw_hyper(lrs[:nt], wds[:nt], betas[0], betas[1], step, self.hyper_host[k][: 1 + 2 * nt])
        ops.adamw_hyper(lrs[nt:], wds[nt:], betas[0], betas[1], step, self.hyper_host[k][1 + 2 * nt:])
        self.hyper_dev.copy_(self.hyper_host[k], non_blocking=True)

    def _slot(self) -> int:
        """pinned staging slot of this step (ring of four; a slot is rewritten only after the copies that read it completed)"""
        k = self._ring
        ev = self._ring_ev[k]
        if ev is not None:
            ev.synchronize()
        return k

    def _slot_done(self):
        k = self._ring
        if self._ring_ev[k] is None:
            self._ring_ev[k] = torch.cuda.Event()
        self._ring_ev[k].record()
        self._ring = (k + 1) % 4

    # ------------------------------------------------------------------------------------------------------------------
    def _sequence(self, pg: _Program):
        """The step as straight-line library calls.  `pg` decides what is captured where.

        Two linear chains.  MAIN (caller's stream): weight prep -> temporal forward -> [text features] -> selector forward ->
        loss -> selector backward -> temporal backward -> AdamW (temporal model).  TEXT stream: text backward -> AdamW (ctx,
        text_projection) -> text forward OF THE NEXT STEP.  The text tower's forward depends on ctx / text_projection only,
        so it is software-pipelined across steps: step k+1's text features are computed at the end of step k, right after the
        text parameters' own AdamW, beside step k's temporal backward -- the 340 latency-bound launches of a rank's text tower
        (1.3 ms at 2 classes) leave the critical path.  Same arithmetic in the same order; a prologue replays the text
        forward at the start of a step whenever the pipelined features are not current (first step, parameters edited)."""
        par = _par()
        mod, net = self.mod, self.net
        sel, tm, te, pl = net.selector_model, net.temporal_model, net.text_encoder, net.prompt_learner
        N, Lg, C1 = sel.num_segments, sel.seg_length, len(sel.classnames) - 1
        B, rows = self.B, self.x.shape[0]
        nc = mod.ncentroid
        dist_on = self.distributed
        world = self.world
        ctx_p, P_p = pl.ctx, te.text_projection
        C_all = pl.n_cls
        cp = bool(getattr(net, "text_class_parallel", True)) and dist_on
        lo, hi = par.shard_range(C_all, world, par.rank()) if cp else (0, C_all)
        concat = bool(net.concat_features)
        tparams = Fn._temporal_param_list(tm)
        x, labels = self.x, self.labels
        mt, mb = self.masks[0], self.masks[1]
        keep = pg.keep
        pipelined = self.in_graph_adamw
        gscale = 1.0 / world if dist_on else 1.0

        # ---- text tower forward (text stream): prologue of a step / tail of the previous one
        tf_full = self.tf_full if cp else None
        tbox = {}
        keep.append(tbox)

        def text_fwd():
            if cp:
                ops.fill_(tf_full, 0.0)
            if hi > lo:
                tbox["tf"], tbox["state"] = Fn._text_forward_rows(net, ctx_p, P_p, lo, hi, tf_out=tf_full[lo:hi] if cp else None)

        def text_fwd_items():
            pg.on_x(text_fwd, after=pg.mark())
            if cp:                                                       # class-parallel exchange of the text features
                pg.x_eager(lambda: par.assemble_rows_into(tf_full, lo, hi))
            pg.x_mark("text_fwd")
        i0 = len(pg.items)
        text_fwd_items()
        self.n_prologue = len(pg.items) - i0
        tf = tf_full if cp else tbox.get("tf")

        # ---- main: derived weight layouts of the temporal model (the optimizer rewrote the masters), temporal forward
        pg.begin()
        tm.refresh_prepared(True)
        keep.append(tm._derived(True))                                   # the captured launches point into these buffers
        cT = SimpleNamespace(needs_input_grad=(concat,))
        cT.grad_out = self.gviews
        scores = None
        if not concat:
            scores = Fn.TemporalFn.forward(cT, x, nc, tm, *tparams).view(-1)
        pg.main_wait("text_fwd")                                         # the text features of this step
        pg.begin()
        # ---- selector forward (selector_model.py:32-99, 119-333)
        dirs = ops.text_directions(tf, nc, sel.normal_id)
        raw, mean, var_b, var_u = ops.selector_project_stats(x, nc, dirs)
        total_rows = rows
        bn = sel.bn_layer
        # the forward tail (BatchNorm, running statistics, picks, gather of the top-k segments) as ONE launch per step when a
        # video's logits fit the LDS (acx_selector_tail: bit-identical to the separate launches below, which the autograd path uses)
        fuse_tail = ((N * Lg * C1 + N * C1 + 2 * N + 2 * C1 + sel.num_topk + sel.num_bottomk) * 4 <= 160 * 1024 and C1 <= 64
                     and not _UNFUSED)
        if dist_on:                                                      # SyncBatchNorm (configs/trainer/ddp.yaml:9)
            ops.bn_pack(mean, var_b, rows, out=self.bn_local)
            pg.eager(lambda: par.all_gather_into(self.bn_gathered, self.bn_local))
            pg.begin()
            if not fuse_tail:
                mean, var_b, var_u, total_rows = ops.bn_combine(self.bn_gathered, C1)
        if fuse_tail:
            logits, idx_top, idx_bot, logits_topk, st = ops.selector_tail(
                raw, labels, mt, mb, N, Lg, sel.normal_id, sel.num_topk, sel.num_bottomk, bn.eps,
                gathered=self.bn_gathered if dist_on else None, stats=None if dist_on else (mean, var_b, var_u), bn=bn)
            if dist_on:
                mean, var_b, var_u, total_rows = st
        else:
            logits = ops.selector_bn(raw, mean, var_b, bn.eps)
            ops.bn_running_update_(bn, mean, var_u)
            idx_top, idx_bot = ops.select_idx(logits, labels, mt, mb, N, Lg, sel.normal_id, sel.num_topk, sel.num_bottomk)
            logits_topk = ops.gather_segments(logits, idx_top, N, Lg)
        half = B // 2
        ia, in_, ba = idx_top[:half], idx_top[half:], idx_bot[:half]
        if concat:
            Kp = tm.prepared(True)["Kp"]
            feats = ops.concat_features(logits, x, nc, Kp)
            scores = Fn.TemporalFn.forward(cT, feats, None, tm, *tparams).view(-1)
        # ---- loss forward + its gradients for an upstream gradient of 1 (loss.py:51-195)
        crit = mod.criterion
        lambdas = (crit.lambda_dir_abn, crit.lambda_dir_nor, crit.lambda_topk_abn, crit.lambda_bottomk_abn,
                   crit.lambda_topk_nor, crit.lambda_smooth, crit.lambda_sparse)
        # loss + `meter += losses` + the scatter of the top-k rows' gradient + the BatchNorm backward sums as ONE launch
        # (acx_mil_loss_bn: bit-identical to the four separate steps) when nothing sits between the loss and the selector's backward
        fuse_loss = (not concat and rows % 256 == 0 and rows <= 131072 and C1 <= 64 and crit.num_segments == N
                     and crit.frames_per_segment == Lg and crit.num_topk == sel.num_topk and not _UNFUSED)
        bn_sums = None
        if fuse_loss:
            losses, dsim, dsc, bn_sums = ops.mil_loss_bn(logits, logits_topk, labels, scores, ia, in_, ba, crit.num_segments,
                                                         crit.frames_per_segment, crit.num_topk, crit.normal_id, lambdas,
                                                         meter=self.meter_sum)
            dtopk = None
        else:
            losses, dsim, dtopk, dsc = ops.mil_loss(logits, logits_topk, labels, scores, ia, in_, ba, crit.num_segments,
                                                    crit.frames_per_segment, crit.num_topk, crit.normal_id, lambdas)
            ops.axpby_(self.meter_sum, losses, 1.0, 1.0)                 # the eight running loss sums (module's meters)
        self.losses = losses
        keep.extend([dsc, cT, scores, logits, x])
        d_scores = dsc.view(-1, 1)
        g_ctx, g_P = self.gviews[ctx_p], self.gviews[P_p]

        def temporal_bwd():
            return Fn.TemporalFn.backward(cT, d_scores)[0]

        def selector_bwd(dl):
            if bn_sums is not None:                                      # (dl already holds the scattered top-k gradient)
                sums = bn_sums
            else:
                ops.scatter_segments_(dl, dtopk, idx_top, N, Lg)
                sums = ops.bn_bwd_stats(logits, dl)
            if dist_on:
                pg.eager(lambda: par.all_reduce_sum_(sums))
                pg.begin()
            draw = ops.bn_bwd_apply(logits, dl, var_b, sums, total_rows, bn.eps)
            d_text = ops.selector_dirs_grad(draw, x, nc, tf, sel.normal_id, C1)
            if cp:
                pg.eager(lambda: par.all_reduce_sum_(d_text))
            return d_text

        def text_bwd(d_text):
            # dX through the frozen tower, both gradients produced in their flat-buffer views
            if hi > lo:
                d_tf = d_text[lo:hi]
                if cp and ctx_p.dim() == 3:
                    ops.fill_(g_ctx, 0.0)
                    Fn._text_backward_rows(net, P_p, tbox["state"], d_tf, d_ctx_out=g_ctx[lo:hi], d_P_out=g_P)
                else:
                    Fn._text_backward_rows(net, P_p, tbox["state"], d_tf, d_ctx_out=g_ctx, d_P_out=g_P)
            else:                                                        # more ranks than classes: zero gradients
                ops.fill_(g_ctx, 0.0)
                ops.fill_(g_P, 0.0)

        if not concat:
            # selector backward first (short): the text stream's work then runs BESIDE the temporal backward, which is enqueued
            # first (a 75-node graph worth 2 ms of GPU time; the text graphs take the host 0.6 ms to enqueue)
            d_text = selector_bwd(dsim)
            tok1 = pg.mark()
            pg.begin()
            temporal_bwd()
        else:
            # the temporal model sits between the selector's forward and backward: d_features flows back into the logits
            d_feats = temporal_bwd()
            dcat = torch.empty(rows, C1, dtype=torch.float32, device=x.device)
            ops.prep_multi([(d_feats, dcat, rows, C1, d_feats.stride(0), C1, 0)])
            dl = ops.add(dsim, dcat)
            d_text = selector_bwd(dl)
            tok1 = pg.mark()
        keep.append(d_text)
        # ---- text stream: text backward, [exchange of the text gradients], AdamW of ctx / text_projection, next step's text forward
        tms, tvs = None, None
        if self.in_graph_adamw:
            ms, vs, lrs, wds, betas, eps = self._opt_state()
            nt = len(self.text_params)
            tms, tvs = ms[:nt], vs[:nt]

        def text_adamw():
            ops.adamw_multi_dev_(self.text_params, [self.gviews[p] for p in self.text_params], tms, tvs, self.hyper_dev[: 1 + 2 * nt],
                                 gscale, betas[0], betas[1], eps)

        if dist_on:
            pg.on_x(lambda: text_bwd(d_text), after=tok1)
            pg.x_eager(lambda: self.buckets.reduce_now(self.text_params))         # their own bucket, summed over ranks
            if self.in_graph_adamw:
                pg.on_x(text_adamw, after=pg.mark())
        elif self.in_graph_adamw:
            pg.on_x(lambda: (text_bwd(d_text), text_adamw()), after=tok1)
        else:
            pg.on_x(lambda: text_bwd(d_text), after=tok1)
        pg.x_mark("text_params")                                         # ctx / text_projection: gradients (and update) final
        if pipelined:
            if pg.mode == "capture":
                # the SAME graph / exchange items = next step's text forward, minus the prologue's leading (record, x_wait) pair:
                # here it follows the text AdamW in the text stream's own order and must NOT wait for the main stream
                pg.items.extend(pg.items[i0 + 2:i0 + self.n_prologue])
            else:
                text_fwd_items()
        # ---- main tail: the temporal model's gradient exchange and optimizer step
        if dist_on:
            if not self.in_graph_adamw:
                # finish(average=True) scales the WHOLE flat buffer, the [text_projection, ctx] bucket included: the text
                # stream's backward and its reduce_now must have completed first (with the in-graph AdamW nothing on the main
                # stream touches that bucket: the 1 / world factor is the update's own gscale)
                pg.main_wait("text_params")
            pg.eager(lambda: (self._ready(list(tparams)), self.buckets.finish(average=not self.in_graph_adamw)))
        else:
            pg.host(lambda: (self._ready(list(tparams) + self.text_params), self.buckets.finish(average=False)))
        if self.in_graph_adamw:
            pg.begin()
            ops.adamw_multi_dev_(self.main_params, [self.gviews[p] for p in self.main_params], ms[nt:], vs[nt:],
                                 self.hyper_dev[1 + 2 * nt:], gscale, betas[0], betas[1], eps)
        pg.main_wait("text_params")      # whoever follows on the main stream sees final text parameters / gradients
        pg.end()

    def _ready(self, params):
        self.buckets.mark_ready(params)

    # ------------------------------------------------------------------------------------------------------------------
    def capture(self):
        """Warm-up runs + capture on the inputs loaded so far (load_inputs first: the sequence runs for real twice).  Leaves
        parameters, optimizer state, BatchNorm statistics and meters exactly as they were."""
        par = _par()
        net = self.net
        C_all = net.prompt_learner.n_cls
        E_txt = net.text_encoder.text_projection.shape[1]
        C1 = len(net.selector_model.classnames) - 1
        dev = self.dev
        self.tf_full = torch.zeros(C_all, E_txt, dtype=torch.float32, device=dev)
        self.bn_local = torch.zeros(2 * C1 + 1, dtype=torch.float32, device=dev)
        self.bn_gathered = torch.zeros(self.world, 2 * C1 + 1, dtype=torch.float32, device=dev)
        # static inputs must hold sane data while the sequence runs for real (warm-up): the caller's first batch is copied in
        # before capture() (see AnomalyCLIPModule.train_batch); a state snapshot makes the dry runs side-effect free
        ops.prime_capture_stream(self.cap, dev)
        ops.prime_capture_stream(self.tstream, dev)
        snap = self._snapshot()
        try:
            for _ in range(2):
                if self.in_graph_adamw:
                    self._refresh_hyper()
                self.buckets.arm()
                self._sequence(_Program("eager", self.cap, self.tstream))
                torch.cuda.synchronize()
            self._restore(snap)
            if self.in_graph_adamw:
                self._refresh_hyper()
            torch.cuda.synchronize()
            pg = _Program("capture", self.cap, self.tstream)
            self.buckets.arm()
            try:
                self._sequence(pg)
            except Exception:
                pg.abort()
                raise
            torch.cuda.synchronize()
            self.program, self._keep = pg.items, pg.keep
            self._text_state = None
        finally:
            self._restore(snap)
            torch.cuda.synchronize()

    def _snapshot(self):
        """everything a dry run of the step mutates: trainable parameters, AdamW moments + step counts, BatchNorm running
        statistics, the meters"""
        bn = self.net.selector_model.bn_layer
        ps = [p.detach().clone() for p in self.opt_params]
        st = []
        for p in self.opt_params:
            s = self.opt.state[p] if self.in_graph_adamw else {}
            st.append((s.get("step", 0), None if "exp_avg" not in s else s["exp_avg"].clone(),
                       None if "exp_avg_sq" not in s else s["exp_avg_sq"].clone()))
        return (ps, st, bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone(), self.meter_sum.clone(),
                ops.WEIGHT_EPOCH[0])

    def _restore(self, snap):
        ps, st, rm, rv, nbt, ms, epoch = snap
        bn = self.net.selector_model.bn_layer
        with torch.no_grad():
            for p, q in zip(self.opt_params, ps):
                p.copy_(q)
            for p, (step, m, v) in zip(self.opt_params, st):
                s = self.opt.state[p] if self.in_graph_adamw else {}
                if s:
                    s["step"] = step
                    if m is not None:
                        s["exp_avg"].copy_(m)
                        s["exp_avg_sq"].copy_(v)
                    else:
                        s["exp_avg"].zero_()
                        s["exp_avg_sq"].zero_()
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            bn.num_batches_tracked.copy_(nbt)
            self.meter_sum.copy_(ms)
        ops.WEIGHT_EPOCH[0] = epoch + 1          # derived-weight caches keyed by the epoch must not survive the dry runs

    # ------------------------------------------------------------------------------------------------------------------
    def load_inputs(self, abatch, nbatch, masks):
        """(features, labels) of the abnormal and the normal half + the two host masks -> the static input buffers"""
        af, al = abatch
        nf, nl = nbatch
        T = self.x.shape[0] // self.B
        D = self.x.shape[1]
        xa, xn = self.x[: self.ha * T], self.x[self.ha * T:]
        xa.copy_(af.reshape(-1, D), non_blocking=True)
        xn.copy_(nf.reshape(-1, D), non_blocking=True)
        self.labels[: self.ha].copy_(al.reshape(-1), non_blocking=True)
        self.labels[self.ha:].copy_(nl.reshape(-1), non_blocking=True)
        mt, mb = masks
        if mt.is_cuda:
            self.masks[0].copy_(mt, non_blocking=True)
            self.masks[1].copy_(mb, non_blocking=True)
        else:
            k = self._slot()
            self.masks_host[k][0].copy_(mt)
            self.masks_host[k][1].copy_(mb)
            self.masks.copy_(self.masks_host[k], non_blocking=True)

    def _text_param_state(self):
        pl, te = self.net.prompt_learner, self.net.text_encoder
        return (ops.WEIGHT_EPOCH[0], pl.ctx._version, te.text_projection._version, pl.ctx.data_ptr(), te.text_projection.data_ptr())

    def step(self):
        """one optimisation step on the loaded inputs; returns the static [8] loss tensor"""
        if self.in_graph_adamw:
            self._refresh_hyper()
        self._slot_done()
        self.buckets.arm()
        main = torch.cuda.current_stream()
        items = self.program
        if self.in_graph_adamw and self._text_state == self._text_param_state():
            items = items[self.n_prologue:]          # this step's text features were computed at the end of the previous step
        skip_x = _SKIP_TEXT
        trace = _HOST_TRACE
        if trace is not None:
            import time as _time
            rec = [("begin", _time.perf_counter())]
        for it in items:
            kind = it[0]
            if trace is not None:
                rec.append((kind + ":" + str(it[2] if kind == "graph" else (it[1] if isinstance(it[1], (int, str)) else "")), _time.perf_counter()))
                if kind == "graph" and it[2] == "main":      # device time stamps in front of every main-stream segment
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(main)
                    rec.append(("ev", ev))
            if skip_x and (kind == "x_eager" or (kind == "graph" and it[2] != "main")):
                continue
            if kind == "graph":
                if it[2] == "main":
                    it[1].replay()
                else:
                    with torch.cuda.stream(self.tstream):
                        it[1].replay()
            elif kind == "eager" or kind == "host":
                it[1]()
            elif kind == "x_eager":
                with torch.cuda.stream(self.tstream):
                    it[1]()
            elif kind == "record":
                self.ev_marks[it[1]].record(main)
            elif kind == "x_wait":
                self.tstream.wait_event(self.ev_marks[it[1]])
            elif kind == "record_x":
                self.ev_x[it[1]].record(self.tstream)
            elif kind == "main_wait_x":
                main.wait_event(self.ev_x[it[1]])
        if trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(main)
            rec.append(("ev", ev))
            rec.append(("end", _time.perf_counter()))
            trace.append(rec)
        if self.in_graph_adamw:
            ops.WEIGHT_EPOCH[0] += 1
            self._text_state = self._text_param_state()
        else:
            self.opt.step()
        return self.losses
