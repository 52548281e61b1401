"""Training-side glue: torch.autograd.Function wrappers whose forward AND backward run libacx
kernels (PyTorch supplies the autograd graph, device memory and streams only).

Three coarse functions mirror the reference's trainable sub-graphs:
  TextFeaturesFn   prompt_learner.ctx, text_encoder.text_projection -> text features
                   (coop.py:74-90, text_encoder.py:14-25; frozen CLIP text layers, dX only)
  SelectorTrainFn  text features -> (logits, logits_topk, logits_bottomk) + MIL indices
                   (selector_model.py:32-99, 119-333)
  TemporalFn       temporal_model.* -> scores (temporal_model.py:42-77 + restated axial transformer)
  MilLossFn        loss.py:51-195 (forward+backward in one kernel pass)
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import _lib as L
from .. import ops


# ------------------------------------------------------------------------------------------------------
def _parallel():
    from .. import parallel
    return parallel


# ====================================================================================================== text path
def _text_forward_rows(net, ctx_param, text_projection, lo: int, hi: int, tf_out=None):
    """Text features of classes [lo, hi) (coop.py:74-90, text_encoder.py:14-25): prompt assembly, the 12 frozen CLIP
    text layers, EOT gather, ln_final, @ text_projection.  Returns (features [hi-lo, E], state for the backward).
    tf_out: [hi-lo, E] destination (rows [lo, hi) of the class-parallel exchange buffer)."""
    te, pl = net.text_encoder, net.prompt_learner
    tr = te.transformer
    W, heads, Lc = tr.width, tr.heads, te.positional_embedding.shape[0]
    shared = ctx_param.dim() == 2
    cp = ctx_param.detach() if shared else ctx_param.detach()[lo:hi]
    # causal tower, EOT rows only: positions behind the last EOT are inert -- the prompt kernel emits the first Le only
    Lc = min(int(getattr(net, "text_len", Lc)), Lc)
    x0 = ops.prompt_embed(pl.token_prefix[lo:hi], cp, pl.token_suffix[lo:hi], te.positional_embedding.detach(), pl.n_ctx, Lout=Lc)
    C = x0.shape[0]
    x = x0.view(C * Lc, W)
    saved = []
    # few rows (a data-parallel rank's block of classes: 77 rows per class): acx_gemm runs the few-row kernel and the
    # activation / its derivative ride in its prologue / epilogue instead of separate launches
    few = C * Lc <= ops.SK_MAX_ROWS and (4 * W) % 256 == 0 and W % 256 == 0
    # ... and the two LayerNorms of a block ride in the A prologue of the GEMM that consumes them (K == 512: the whole row is
    # resident in the kernel's two stages; LN(x) is never materialised -- the backward recomputes it from x anyway)
    norm_fused = few and W == 512
    for blk in tr.resblocks:
        if norm_fused:
            qkv = ops.gemm(x, blk.attn.in_proj_weight.detach(), bias=blk.attn.in_proj_bias.detach(),
                           a_norm=(blk.ln_1.weight.detach(), blk.ln_1.bias.detach()))
        else:
            h1 = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias)
            qkv = ops.gemm(h1, blk.attn.in_proj_weight.detach(), bias=blk.attn.in_proj_bias.detach())
        att = ops.attention(qkv, C, Lc, heads, True)
        x_mid = ops.gemm(att, blk.attn.out_proj.weight.detach(), bias=blk.attn.out_proj.bias.detach(), residual=x)
        if norm_fused:
            pre = ops.gemm(x_mid, blk.mlp.c_fc.weight.detach(), bias=blk.mlp.c_fc.bias.detach(),
                           a_norm=(blk.ln_2.weight.detach(), blk.ln_2.bias.detach()))
        else:
            h2 = ops.layernorm(x_mid, blk.ln_2.weight, blk.ln_2.bias)
            pre = ops.gemm(h2, blk.mlp.c_fc.weight.detach(), bias=blk.mlp.c_fc.bias.detach())
        if few:                                       # QuickGELU applied as the few-row GEMM reads its A operand
            x_next = ops.gemm(pre, blk.mlp.c_proj.weight.detach(), bias=blk.mlp.c_proj.bias.detach(), residual=x_mid,
                              a_act=L.ACT_QUICKGELU)
        else:
            act = ops.act(pre, None, 2)
            x_next = ops.gemm(act, blk.mlp.c_proj.weight.detach(), bias=blk.mlp.c_proj.bias.detach(), residual=x_mid)
        saved.append((x, qkv, x_mid, pre))
        x = x_next
    rows_idx = _eot_rows(net, lo, hi, Lc)
    eot = ops.gather_rows(x, rows_idx)
    eln = ops.layernorm(eot, te.ln_final.weight, te.ln_final.bias)
    tf = ops.gemm(eln, ops.transpose(text_projection.detach()), out=tf_out)
    return tf, (saved, rows_idx, eot, eln, C, Lc, W, heads, shared)


def _eot_rows(net, lo, hi, Lc):
    """row index of every local class's EOT token in the [C_loc * Lc, W] activation matrix; built once per (block, length)"""
    cache = net.__dict__.setdefault("_eot_rows_cache", {})
    key = (lo, hi, Lc, net.eot_index.data_ptr(), net.eot_index._version)   # in-place edits (load_state_dict) bump _version
    r = cache.get(key)
    if r is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.AcxError("text tower: the EOT row table must be built by an eager call before a HIP graph is captured")
        dev = net.eot_index.device
        r = cache[key] = (torch.arange(hi - lo, device=dev, dtype=torch.int64) * Lc + net.eot_index[lo:hi]).contiguous()
    return r


def _text_backward_rows(net, text_projection, state, d_tf, d_ctx_out=None, d_P_out=None):
    """d(features of the local classes) -> (d ctx for those classes [C_loc, n_ctx, W] (or [n_ctx, W] when shared),
    d text_projection [W, E]); dX only through the frozen layers.  d_ctx_out / d_P_out: destinations to produce the two
    gradients in (views of parallel.GradBuckets.flat)."""
    te, pl = net.text_encoder, net.prompt_learner
    saved, rows_idx, eot, eln, C, Lc, W, heads, shared = state
    d_tf = d_tf.contiguous()
    d_P = ops.gemm_tn(eln, d_tf, out=d_P_out)                                # [W, E]
    d_eln = ops.gemm(d_tf, text_projection.detach().contiguous())            # d_tf @ P^T  (W_op = P [W,E])
    d_eot, _, _ = ops.layernorm_bwd(eot, te.ln_final.weight, d_eln, need_params=False)
    d_x = ops.scatter_rows(d_eot, rows_idx, C * Lc)
    wt = _frozen_transposes(te)
    few = C * Lc <= ops.SK_MAX_ROWS and (4 * W) % 256 == 0 and W % 256 == 0
    for li in range(len(saved) - 1, -1, -1):
        blk = te.transformer.resblocks[li]
        x, qkv, x_mid, pre = saved[li]
        t_in, t_out, t_fc, t_proj = wt[li]
        if few:
            d_pre = ops.gemm(d_x, t_proj, gelu_grad_of=pre)                  # (d_x @ proj_w) * gelu'(pre) in the epilogue
        else:
            d_act = ops.gemm(d_x, t_proj)                                    # [rows,4W] = d_x @ proj_w
            d_pre = ops.act(pre, d_act, 1)
        d_h2 = ops.gemm(d_pre, t_fc)                                         # [rows,W]
        d_x_mid, _, _ = ops.layernorm_bwd(x_mid, blk.ln_2.weight, d_h2, need_params=False, add=d_x)   # d_x + LN2'
        d_att = ops.gemm(d_x_mid, t_out)
        d_qkv = ops.seq_attention_bwd(qkv, d_att, C, 1, Lc, heads, 64, 1, causal=True)
        d_h1 = ops.gemm(d_qkv, t_in)
        d_x, _, _ = ops.layernorm_bwd(x, blk.ln_1.weight, d_h1, need_params=False, add=d_x_mid)      # d_x_mid + LN1'
    d_ctx = ops.ctx_grad(d_x, C, pl.n_ctx, Lc, W, shared, out=d_ctx_out)
    return d_ctx, d_P


class TextFeaturesFn(torch.autograd.Function):
    """Text features of ALL classes on this rank, or -- class-parallel under data parallelism -- of this rank's
    contiguous block of classes followed by an exchange (the text encoder is the one part
    of the step that plain DP replicates; 83 GFLOP forward + dX backward at 14 classes do not shrink with more GPUs).

    class_parallel: forward  = local rows written into a zero (C, E) buffer, ONE all-reduce (sum with zeros is exact);
                    backward = ONE all-reduce of d_features (every rank holds the gradient of ITS videos w.r.t. ALL
                    classes), then the local classes' rows go through the local backward.  The returned d_ctx is
                    non-zero only in the local class block and d_text_projection is the local classes' partial sum --
                    both already summed over ranks' videos -- so the gradient all-reduce (sum / world) that follows
                    (parallel.GradBuckets) yields exactly the data-parallel mean the reference's DDP produces."""

    @staticmethod
    def forward(ctx, ctx_param, text_projection, net, class_parallel):
        par = _parallel()
        C_all = net.prompt_learner.n_cls
        cp = bool(class_parallel) and par.is_distributed()
        lo, hi = par.shard_range(C_all, par.world_size(), par.rank()) if cp else (0, C_all)
        if hi == lo:
            # more ranks than classes (XD-Violence: 7 classes on 8 GPUs): this rank owns no class.  It contributes an empty
            # block and zero parameter gradients but still joins BOTH exchanges -- the other ranks block in them
            tf_loc, state = text_projection.new_zeros((0, text_projection.shape[1])), None
        else:
            tf_loc, state = _text_forward_rows(net, ctx_param, text_projection, lo, hi)
        if cp:
            tf = par.assemble_rows(tf_loc, lo, C_all)
        else:
            tf = tf_loc
        ctx.net, ctx.state, ctx.rows, ctx.cp = net, state, (lo, hi, C_all), cp
        ctx.save_for_backward(text_projection, ctx_param)
        return tf

    @staticmethod
    def backward(ctx, d_tf):
        net = ctx.net
        lo, hi, C_all = ctx.rows
        P, ctx_param = ctx.saved_tensors
        d_tf = d_tf.contiguous()
        if ctx.cp:
            d_tf = _parallel().all_reduce_sum_(d_tf.clone())[lo:hi]
        if hi == lo:
            return torch.zeros_like(ctx_param), torch.zeros_like(P), None, None
        d_ctx_loc, d_P = _text_backward_rows(net, P, ctx.state, d_tf)
        if ctx.cp and ctx_param.dim() == 3:
            d_ctx = torch.zeros_like(ctx_param)
            d_ctx[lo:hi] = d_ctx_loc
        else:
            d_ctx = d_ctx_loc
        ctx.state = None
        return d_ctx, d_P, None, None


def _frozen_transposes(te):
    """Transposed copies of the FROZEN text-transformer weights (dX = dY @ W needs W^T as the [N,K] operand)."""
    # optimizer steps through raw pointers (ops.adamw_) do not bump _version, hence WEIGHT_EPOCH -- but only for weights an
    # optimizer can touch: the CLIP text layers are frozen (requires_grad False), their transposes survive the step
    frozen = not any(p.requires_grad for p in te.transformer.parameters())
    key = ((-1 if frozen else ops.WEIGHT_EPOCH[0]),) + tuple((p.data_ptr(), p._version) for p in te.transformer.parameters())
    cache = getattr(te, "_wt_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    out = []
    for blk in te.transformer.resblocks:
        out.append((ops.transpose(blk.attn.in_proj_weight.detach()), ops.transpose(blk.attn.out_proj.weight.detach()),
                    ops.transpose(blk.mlp.c_fc.weight.detach()), ops.transpose(blk.mlp.c_proj.weight.detach())))
    te._wt_cache = (key, out)
    return out


def text_features_train(net):
    return TextFeaturesFn.apply(net.prompt_learner.ctx, net.text_encoder.text_projection, net,
                                getattr(net, "text_class_parallel", True))


# ------------------------------------------------------------------------------------------------------ graph / side stream
# `net.text_graph = True`: the text tower of a training step -- ~100 launches forward, ~130 backward, all of them small
# (77 C rows) and the same every step -- is captured once into two HIP graphs and replayed on a SIDE stream:
#   * the host issues 2 launches instead of ~230 library calls (the step is launch-bound from 16 videos per GPU down);
#   * the tower runs next to the temporal model instead of in front of it: the forward graph is launched first and
#     joined only where the selector needs the text features (configs without the logits concat run the temporal model
#     in between); the backward graph waits for the selector's backward only and runs beside the temporal backward.
# Same kernels in the same order on the same operands as the eager path: results are bit-identical (tests).
class _TextGraphs:
    empty = False
    # capture_error_mode="thread_local": other threads of the process (the RCCL watchdog polling its events) must not
    # invalidate a capture in progress
    def __init__(self, net, lo, hi):
        te, pl = net.text_encoder, net.prompt_learner
        self.key = self.make_key(net, lo, hi)
        self.side = torch.cuda.Stream()
        ctx_param, P = pl.ctx, te.text_projection
        _frozen_transposes(te)                                               # cached outside the graphs
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            for _ in range(2):                                               # warm-up: lazy workspaces, function attributes
                tf, state = _text_forward_rows(net, ctx_param, P, lo, hi)
                _text_backward_rows(net, P, state, torch.zeros_like(tf))
        torch.cuda.synchronize()
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, stream=self.side, capture_error_mode="thread_local"):
            self.tf, self.state = _text_forward_rows(net, ctx_param, P, lo, hi)
        self.d_tf = torch.zeros_like(self.tf)
        self.g_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_bwd, stream=self.side, pool=self.g_fwd.pool(), capture_error_mode="thread_local"):
            self.d_ctx, self.d_P = _text_backward_rows(net, P, self.state, self.d_tf)
        torch.cuda.synchronize()
        self.fwd_done = torch.cuda.Event()

    @staticmethod
    def make_key(net, lo, hi):
        te, pl = net.text_encoder, net.prompt_learner
        ts = (pl.ctx, pl.token_prefix, pl.token_suffix, te.text_projection, te.positional_embedding, net.eot_index)
        # the frozen layers' parameter OBJECTS are listed once (Module.parameters() walks the module tree: 0.4 ms per step
        # for 144 tensors); their addresses are still read every step
        plist = te.__dict__.get("_acx_plist")
        if plist is None:
            plist = te.__dict__["_acx_plist"] = list(te.transformer.parameters())
        return (lo, hi, int(getattr(net, "text_len", 0)), net.eot_index._version) + tuple((t.data_ptr(), tuple(t.shape)) for t in ts) + \
            tuple(p.data_ptr() for p in plist)


class _NoTextRows:
    """Stand-in for _TextGraphs on a rank that owns no class (more ranks than classes): nothing to replay; an empty
    feature block forward, zero parameter gradients backward, and the same exchanges as every other rank."""
    empty = True

    def __init__(self, net, lo, hi):
        P = net.text_encoder.text_projection
        self.key = _TextGraphs.make_key(net, lo, hi)
        self.tf = P.new_zeros((0, P.shape[1]))
        self.fwd_done = torch.cuda.Event()


def text_graph_launch(net):
    """Start this step's text forward on the side stream (no autograd node yet: TextGraphFn picks the result up)."""
    par = _parallel()
    C_all = net.prompt_learner.n_cls
    cp = bool(getattr(net, "text_class_parallel", True)) and par.is_distributed()
    lo, hi = par.shard_range(C_all, par.world_size(), par.rank()) if cp else (0, C_all)
    tg = getattr(net, "_text_graphs", None)
    if hi == lo:
        if tg is None or tg.key != _TextGraphs.make_key(net, lo, hi):
            tg = net._text_graphs = _NoTextRows(net, lo, hi)
        tg.rows, tg.cp = (lo, hi, C_all), cp
        tg.fwd_done.record()
        return tg
    if tg is None or tg.key != _TextGraphs.make_key(net, lo, hi):
        tg = net._text_graphs = _TextGraphs(net, lo, hi)
    tg.rows, tg.cp = (lo, hi, C_all), cp
    tg.side.wait_stream(torch.cuda.current_stream())                         # parameters of the last optimizer step
    with torch.cuda.stream(tg.side):
        tg.g_fwd.replay()
        tg.fwd_done.record()
    return tg


class TextGraphFn(torch.autograd.Function):
    """Autograd node of the graph-replayed text tower; applied BEFORE the temporal model so that its backward is the
    LAST node of the step (autograd runs later-created nodes first): by then the temporal backward is queued on the main
    stream and the text backward, which waits only for the event recorded after the selector's backward, runs beside it."""

    @staticmethod
    def forward(ctx, ctx_param, text_projection, net):
        ctx.net = net
        ctx.save_for_backward(ctx_param)
        return net._text_graphs.tf.detach()             # alias of the graph's static output; valid after text_graph_join()

    @staticmethod
    def backward(ctx, d_tf_loc):
        net = ctx.net
        tg = net._text_graphs
        (ctx_param,) = ctx.saved_tensors
        lo, hi, C_all = tg.rows
        ev = getattr(net, "_text_grad_ready", None)
        net._text_grad_ready = None
        if getattr(tg, "empty", False):
            return torch.zeros_like(ctx_param), torch.zeros_like(net.text_encoder.text_projection), None
        if ev is not None:
            tg.side.wait_event(ev)                      # d_tf is final: recorded after the selector / assemble backward
        else:
            tg.side.wait_stream(torch.cuda.current_stream())
        d_tf_loc.record_stream(tg.side)                 # allocated on the main stream, read by the side stream's copy
        with torch.cuda.stream(tg.side):
            tg.d_tf.copy_(d_tf_loc)
            tg.g_bwd.replay()
        torch.cuda.current_stream().wait_stream(tg.side)
        d_ctx_loc, d_P = tg.d_ctx, tg.d_P
        if tg.cp and ctx_param.dim() == 3:
            d_ctx = torch.zeros_like(ctx_param)
            d_ctx[lo:hi] = d_ctx_loc
        else:
            d_ctx = d_ctx_loc.clone()
        return d_ctx, d_P.clone(), None


class TextAssembleFn(torch.autograd.Function):
    """Class-parallel exchange around the graph path (what TextFeaturesFn does inline): forward = local rows into a zero
    (C, E) buffer + ONE all-reduce; backward = ONE all-reduce of d_features, this rank's class rows."""

    @staticmethod
    def forward(ctx, tf_loc, net):
        tg = net._text_graphs
        lo, hi, C_all = tg.rows
        ctx.net = net
        return _parallel().assemble_rows(tf_loc, lo, C_all) if tg.cp else tf_loc.clone()

    @staticmethod
    def backward(ctx, d_tf):
        net = ctx.net
        tg = net._text_graphs
        lo, hi, _ = tg.rows
        d_tf = d_tf.contiguous()
        if tg.cp:
            d_tf = _parallel().all_reduce_sum_(d_tf.clone())[lo:hi].contiguous()
        ev = torch.cuda.Event()
        ev.record()                                     # the text backward may start here
        net._text_grad_ready = ev
        return d_tf, None


def text_graph_join(net, tf_loc):
    """Main stream waits for the forward graph; returns the (C, E) text features."""
    torch.cuda.current_stream().wait_event(net._text_graphs.fwd_done)
    return TextAssembleFn.apply(tf_loc, net)


# ====================================================================================================== selector
class SelectorTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text_features, x, labels, ncentroid, mask_top, mask_bot, sel):
        N, Lg, C1 = sel.num_segments, sel.seg_length, len(sel.classnames) - 1
        x = x.contiguous()
        tf = text_features.detach().contiguous()
        dirs = ops.text_directions(tf, ncentroid, sel.normal_id)
        raw, mean, var_b, var_u = ops.selector_project_stats(x, ncentroid, dirs)     # statistics in the projection's epilogue
        par = _parallel()
        rows = raw.shape[0]
        total_rows = rows
        if par.is_distributed():                                                # SyncBatchNorm (configs/trainer/ddp.yaml:9)
            mean, var_b, var_u, total_rows = par.sync_bn_stats(mean, var_b, rows)
        logits = ops.selector_bn(raw, mean, var_b, sel.bn_layer.eps)
        bn = sel.bn_layer
        ops.axpby_(bn.running_mean, mean, bn.momentum, 1.0 - bn.momentum)        # BatchNorm1d running stats
        ops.axpby_(bn.running_var, var_u, bn.momentum, 1.0 - bn.momentum)
        bn.num_batches_tracked += 1
        B = labels.shape[0]
        labels_d = labels.to(device=x.device, dtype=torch.int64).contiguous()
        mt = mask_top.to(device=x.device, dtype=torch.float32).contiguous()
        mb = mask_bot.to(device=x.device, dtype=torch.float32).contiguous()
        idx_top, idx_bot = ops.select_idx(logits, labels_d, mt, mb, N, Lg, sel.normal_id, sel.num_topk, sel.num_bottomk)
        logits_topk = ops.gather_segments(logits, idx_top, N, Lg)
        logits_bottomk = ops.gather_segments(logits, idx_bot, N, Lg)
        ctx.sel, ctx.dims = sel, (N, Lg, C1, total_rows)
        ctx.save_for_backward(tf, x, ncentroid, logits, var_b, idx_top, idx_bot)
        ctx.mark_non_differentiable(idx_top, idx_bot)
        return logits, logits_topk, logits_bottomk, idx_top, idx_bot

    @staticmethod
    def backward(ctx, d_logits, d_topk, d_bottomk, _a, _b):
        sel = ctx.sel
        N, Lg, C1, total_rows = ctx.dims
        tf, x, nc, logits, var_b, idx_top, idx_bot = ctx.saved_tensors
        dl = d_logits.contiguous().clone() if d_logits is not None else torch.zeros_like(logits)
        if d_topk is not None:
            ops.scatter_segments_(dl, d_topk.contiguous(), idx_top, N, Lg)
        if d_bottomk is not None:
            ops.scatter_segments_(dl, d_bottomk.contiguous(), idx_bot, N, Lg)
        sums = ops.bn_bwd_stats(logits, dl)
        par = _parallel()
        if par.is_distributed():
            par.all_reduce_sum_(sums)
        draw = ops.bn_bwd_apply(logits, dl, var_b, sums, total_rows, sel.bn_layer.eps)     # [rows, C1 padded to 4]
        d_text = ops.selector_dirs_grad(draw, x, nc, tf, sel.normal_id, C1)                # d_dirs = draw^T (x - nc), then dirs' backward
        return d_text, None, None, None, None, None, None


def selector_train(sel, x, text_features, labels, ncentroid, masks):
    logits, lt, lb, idx_top, idx_bot = SelectorTrainFn.apply(text_features, x, labels, ncentroid, masks[0], masks[1], sel)
    half = idx_top.shape[0] // 2
    return logits, lt, lb, idx_top[:half], idx_top[half:], idx_bot[:half]


# ====================================================================================================== temporal
def _temporal_param_list(tm) -> List[torch.nn.Parameter]:
    """tm.parameters() in registration order, listed once per module (the tree walk costs ~0.2 ms per call)."""
    plist = tm.__dict__.get("_acx_plist")
    if plist is None:
        plist = tm.__dict__["_acx_plist"] = list(tm.parameters())
    return plist


class TemporalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, a_sub, tm, *params):
        P = tm.prepared(True)
        N, Lg, E, depth = tm.num_segments, tm.seg_length, tm.emb_size, tm.depth
        heads, e = tm.heads, tm.axial_attn.e
        x = feats.reshape(-1, feats.shape[-1]).contiguous()
        tiles = x.shape[0] // (N * Lg)
        x0 = ops.gemm(x, P["proj_w"], bias=tm.projection.bias.detach(), a_sub=a_sub, gn=N, gl=Lg, seg=1,
                      pos0=P["pos0"], pos1=P["pos1"])
        blks = tm.axial_attn.layers.blocks
        saved = []
        tap = tm.__dict__.get("_act_tap")            # tests only: a dict that receives the conv hidden activations

        def attn(x_in, resid, d, fg, axis):
            pn = getattr(blks[2 * d], fg).net.fn
            h = ops.layernorm(x_in, pn.norm.weight, pn.norm.bias)
            qkv = ops.gemm(h, P[f"qkv_w{d}{fg}"])
            o = ops.axial_attention(qkv, tiles, N, Lg, heads, e, axis)
            out = ops.gemm(o, pn.fn.to_out.weight.detach(), bias=pn.fn.to_out.bias.detach(), residual=resid)
            saved.append(("attn", d, fg, axis, x_in, h, qkv, o))
            return out

        x6 = f"c1_w0f_3" in P                          # the convolutions as bf16 x 6 products (TemporalModel.x6_convs)

        def ff(x_in, resid, d, fg):
            f = getattr(blks[2 * d + 1], fg).net
            if x6:                                       # activations as three bf16 planes: [3, rows, C]
                h = ops.layernorm(x_in, P[f"g{d}{fg}"], P[f"b{d}{fg}"], mode=L.NORM_CHAN, planes_out=True)
                u = ops.gemm_x6(h, P[f"c1_w{d}{fg}_3"], bias=f[1].bias.detach(), act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3, gn=N, gl=Lg,
                                cin=E, planes_out=True)
                out = ops.gemm_x6(u, P[f"c2_w{d}{fg}_3"], bias=f[3].bias.detach(), residual=resid, amap=L.AMAP_CONV3X3, gn=N, gl=Lg,
                                  cin=4 * E)
                saved.append(("ff", d, fg, 0, x_in, h, u, None))
                if tap is not None:
                    tap[(d, fg)] = u[0]                  # the hi plane carries the sign
                return out
            h = ops.layernorm(x_in, P[f"g{d}{fg}"], P[f"b{d}{fg}"], mode=L.NORM_CHAN)
            u = ops.gemm(h, P[f"c1_w{d}{fg}"], bias=f[1].bias.detach(), act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=E)
            out = ops.gemm(u, P[f"c2_w{d}{fg}"], bias=f[3].bias.detach(), residual=resid, amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=4 * E)
            saved.append(("ff", d, fg, 0, x_in, h, u, None))
            if tap is not None:                                  # tests: the hidden activation (its sign = the LeakyReLU side taken)
                tap[(d, fg)] = u
            return out

        x1 = x2 = x0
        for d in range(depth):
            y1 = attn(x2, x1, d, "f", 0)
            y2 = attn(y1, x2, d, "g", 1)
            x1 = ff(y2, y1, d, "f")
            x2 = ff(x1, y2, d, "g")
        c = tm.classifier
        scores = ops.cls_head(x1, x2, c.layer_norm.weight, c.layer_norm.bias, c.linear.weight, c.linear.bias, N, Lg, 0)
        ctx.tm, ctx.saved_acts = tm, saved
        ctx.misc = (x, a_sub, x1, x2, scores, tiles, ctx.needs_input_grad[0])
        return scores.view(-1, 1)

    @staticmethod
    def backward(ctx, d_scores):
        """Returns (d_feats, None, None, *parameter gradients).  `ctx.grad_out` (optional; set by the step graph): a dict
        parameter -> its dense gradient view in parallel.GradBuckets.flat -- the large gradients are then PRODUCED there (the
        weight-gradient GEMMs write the view), the small ones are copied there by one acx_multi_copy launch, and None is
        returned in their place."""
        tm = ctx.tm
        x, a_sub, x1, x2, scores, tiles, need_dfeats = ctx.misc
        N, Lg, E = tm.num_segments, tm.seg_length, tm.emb_size
        heads, e = tm.heads, tm.axial_attn.e
        blks = tm.axial_attn.layers.blocks
        P = tm.prepared(True)
        gout = getattr(ctx, "grad_out", None)
        grads = {}
        c = tm.classifier

        def dest(p, shape2d):
            """the [rows, cols] kernel-layout destination inside p's flat-buffer gradient view, or None"""
            if gout is None or p not in gout:
                return None
            v = gout[p]
            if p.dim() == 4:                                   # conv weight: channels-last view -> [Cout, 9 Cin]
                v = v.permute(0, 2, 3, 1)
                if not v.is_contiguous():
                    return None
            elif not v.is_contiguous():
                return None
            return v.reshape(shape2d)

        def put(p, g, direct):
            grads[p] = None if direct is not None else g

        # Parameter gradients that are NOT fat -- the small weight-gradient GEMMs (to_out, to_q|to_kv, projection), the bias column
        # sums, the LayerNorm / classifier parameter reductions -- are only RECORDED while the dX chain runs and issued at the end
        # as three grouped launches (acx_gemm_tn_group / acx_colsum_fused_group / acx_reduce_rows_group) instead of ~30 latency-
        # bound ones; each group member computes exactly what its single launch computes.
        tn_jobs, cs_jobs, rr_jobs = [], [], []

        dz, part = ops.cls_head_bwd_parts(x1, x2, c.layer_norm.weight, c.layer_norm.bias, c.linear.weight, scores,
                                          d_scores.contiguous().view(-1))

        def cls_params(sred):
            grads[c.layer_norm.weight], grads[c.layer_norm.bias] = sred[:E], sred[E:2 * E]
            grads[c.linear.weight], grads[c.linear.bias] = sred[2 * E:3 * E].view(1, -1), sred[3 * E:3 * E + 1]
        rr_jobs.append((part, cls_params))
        d1, d2 = dz, dz                                   # d(x1), d(x2) of the last block pair

        # every branch ends in a LayerNorm backward: the gradient of the residual path it joins (`add`) is summed in that
        # same pass instead of a separate elementwise launch
        def attn_bwd(rec, d_out, add):
            _, d, fg, axis, x_in, h, qkv, o = rec
            pn = getattr(blks[2 * d], fg).net.fn
            sa = pn.fn
            He = heads * e
            w_dst = dest(sa.to_out.weight, (E, He))
            tn_jobs.append((d_out, o, w_dst, None, lambda g, w=sa.to_out.weight, dd=w_dst: put(w, g, dd)))     # [E, He]
            cs_jobs.append((d_out, lambda g, b=sa.to_out.bias: grads.__setitem__(b, g)))
            d_o = ops.gemm(d_out, P[f"out_wT{d}{fg}"])                                   # [rows, He]
            d_qkv = ops.seq_attention_bwd(qkv, d_o, tiles, N, Lg, heads, e, axis)

            def qkv_params(g_qkv, sa=sa, He=He):                                         # [3He, E]
                grads[sa.to_q.weight], grads[sa.to_kv.weight] = g_qkv[:He], g_qkv[He:]
            tn_jobs.append((d_qkv, h, None, None, qkv_params))
            d_h = ops.gemm(d_qkv, P[f"qkv_wT{d}{fg}"])                                   # [rows, E]
            d_in, lpart = ops.layernorm_bwd_parts(x_in, pn.norm.weight, d_h, add=add)

            def ln_params(sred, pn=pn):
                grads[pn.norm.weight], grads[pn.norm.bias] = sred[:E], sred[E:]
            rr_jobs.append((lpart, ln_params))
            return d_in

        def ff_bwd(rec, d_out, add):
            _, d, fg, _, x_in, h, u, _ = rec
            f = getattr(blks[2 * d + 1], fg).net
            w2_dst, w1_dst = dest(f[3].weight, (E, 36 * E)), dest(f[1].weight, (4 * E, 9 * E))
            cs_jobs.append((d_out, lambda g, b=f[3].bias: grads.__setitem__(b, g)))
            if u.dim() == 3:                             # bf16 x 6 path: h, u are planes; dY goes to planes for its two products
                d_out3 = ops.split_bf16x3(d_out)
                gw2 = ops.gemm_tn_x6(d_out3, u, conv=True, gn=N, gl=Lg, cin=4 * E, out=w2_dst)
                put(f[3].weight, gw2.view(E, 3, 3, 4 * E).permute(0, 3, 1, 2), w2_dst)
                d_u = ops.gemm_x6(d_out3, P[f"c2_dx{d}{fg}_3"], amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=E)
                d_pre, d_pre3 = ops.leaky_grad_planes(u, d_u)                          # LeakyReLU': f32 (bias gradient) + planes
                cs_jobs.append((d_pre, lambda g, b=f[1].bias: grads.__setitem__(b, g)))
                gw1 = ops.gemm_tn_x6(d_pre3, h, conv=True, gn=N, gl=Lg, cin=E, out=w1_dst)
                put(f[1].weight, gw1.view(4 * E, 3, 3, E).permute(0, 3, 1, 2), w1_dst)
                d_h = ops.gemm_x6(d_pre3, P[f"c1_dx{d}{fg}_3"], amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=4 * E)
            else:
                gw2 = ops.gemm_tn(d_out, u, conv=True, gn=N, gl=Lg, cin=4 * E, out=w2_dst)        # [E, 9*4E]  ([Cout][tap][Cin])
                put(f[3].weight, gw2.view(E, 3, 3, 4 * E).permute(0, 3, 1, 2), w2_dst)
                d_u = ops.gemm(d_out, P[f"c2_dx{d}{fg}"], amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=E)
                d_pre = ops.act(u, d_u, 0)                                                 # LeakyReLU'
                cs_jobs.append((d_pre, lambda g, b=f[1].bias: grads.__setitem__(b, g)))
                gw1 = ops.gemm_tn(d_pre, h, conv=True, gn=N, gl=Lg, cin=E, out=w1_dst)            # [4E, 9E]
                put(f[1].weight, gw1.view(4 * E, 3, 3, E).permute(0, 3, 1, 2), w1_dst)
                d_h = ops.gemm(d_pre, P[f"c1_dx{d}{fg}"], amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=4 * E)
            d_in, lpart = ops.layernorm_bwd_parts(x_in, P[f"g{d}{fg}"], d_h, mode=L.NORM_CHAN, add=add)

            def ln_params(sred, f=f):
                grads[f[0].g], grads[f[0].b] = sred[:E].view(1, -1, 1, 1), sred[E:].view(1, -1, 1, 1)
            rr_jobs.append((lpart, ln_params))
            return d_in

        recs = ctx.saved_acts
        for d in range(tm.depth - 1, -1, -1):
            r_af, r_ag, r_ff, r_fg = recs[4 * d: 4 * d + 4]
            # x2' = y2 + ffG(x1')  ;  x1' = y1 + ffF(y2)
            d_y2 = d2
            d1 = ff_bwd(r_fg, d2, d1)                      # d1 + dffG/dx1'
            d_y1 = d1
            d_y2 = ff_bwd(r_ff, d1, d_y2)                  # d_y2 + dffF/dy2
            # y2 = x2 + attnG(y1) ; y1 = x1 + attnF(x2)
            d_x2 = d_y2
            d_y1 = attn_bwd(r_ag, d_y2, d_y1)              # d_y1 + dattnG/dy1
            d_x1 = d_y1
            d_x2 = attn_bwd(r_af, d_y1, d_x2)              # d_x2 + dattnF/dx2
            d1, d2 = d_x1, d_x2
        d_x0 = ops.add(d1, d2)
        pe = tm.axial_attn.pos_emb
        K = tm.input_size
        pw_dst = dest(tm.projection.weight, (E, K)) if P["Kp"] == K else None
        g0, g1 = ops.pos_grad(d_x0, tiles, N, Lg)
        grads[pe.param_0] = ops.transpose(g0).view(1, E, N, 1)         # [N, E] -> (1, E, N, 1), a libacx launch (no torch copy)
        grads[pe.param_1] = ops.transpose(g1).view(1, E, 1, Lg)
        tn_jobs.append((d_x0, x, pw_dst, a_sub, lambda gw: put(tm.projection.weight, gw[:, :K], pw_dst)))        # [E, Kp]
        cs_jobs.append((d_x0, lambda g: grads.__setitem__(tm.projection.bias, g)))
        d_feats = None
        if need_dfeats:
            d_feats = ops.gemm(d_x0, P["proj_wT"])                                       # [rows, Kp]
        # ---- the deferred parameter gradients: three grouped launches (+ one grouped split reduce)
        for job, res in zip(tn_jobs, ops.gemm_tn_group([(a, b, o, bs) for a, b, o, bs, _ in tn_jobs])):
            job[4](res)
        for job, res in zip(cs_jobs, ops.colsum_group([xx for xx, _ in cs_jobs])):
            job[1](res)
        for job, res in zip(rr_jobs, ops.reduce_rows_group([pp for pp, _ in rr_jobs])):
            job[1](res)
        ctx.saved_acts = None
        plist = _temporal_param_list(tm)
        if gout is not None:
            # the gradients that were not produced in place: ONE copy launch into their flat-buffer views
            ys, xs = [], []
            for p in plist:
                g = grads.get(p)
                if g is None:
                    continue
                if p not in gout:
                    raise L.AcxError("TemporalFn.backward: grad_out lacks a parameter's gradient view")
                ys.append(gout[p].reshape(-1))
                xs.append(g.contiguous().reshape(-1))
            ops.multi_copy_(ys, xs)
            return (d_feats, None, None) + (None,) * len(plist)
        out = [grads.get(p) for p in plist]
        # contiguous in the PARAMETER's memory format (conv weights: channels-last, which their [Cout][tap][Cin] gradient
        # views already are -- no copy)
        out = [None if g is None else (g.contiguous() if p.is_contiguous() else
                                       g.contiguous(memory_format=torch.channels_last)) for p, g in zip(plist, out)]
        return (d_feats, None, None, *out)


class _TemporalGraphs:
    """TemporalFn.forward / .backward -- fixed sequences of library calls for a given input shape -- captured once as two
    HIP graphs and replayed on the caller's stream (the autograd path's accelerator; train_batch's whole-step graph,
    step_graph.TrainStepGraph, supersedes it when it can run).  Static buffers: a copy of the input features, the upstream
    gradient, every activation, every returned gradient.  The derived weight layouts (TemporalModel.refresh_prepared: one
    launch) are rebuilt INSIDE the forward graph -- the optimizer rewrites the weights in place between replays."""

    def __init__(self, tm, feats, a_sub, need_dfeats):
        from types import SimpleNamespace
        self.key = self.make_key(tm, feats, a_sub, need_dfeats)
        params = _temporal_param_list(tm)
        self.x = feats.detach().clone()
        self.a_sub = a_sub
        self._P = tm._derived(True)                  # the captured launches point into the derived weight buffers: keep them
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                                               # warm-up: lazy workspaces, function attributes
                c = SimpleNamespace(needs_input_grad=(need_dfeats,))
                tm.refresh_prepared(True)
                sc = TemporalFn.forward(c, self.x, a_sub, tm, *params)
                TemporalFn.backward(c, torch.zeros_like(sc))
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        ops.prime_capture_stream(cap, self.x.device)
        self.cx = SimpleNamespace(needs_input_grad=(need_dfeats,))
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, stream=cap, capture_error_mode="thread_local"), torch.no_grad():
            tm.refresh_prepared(True)
            self.scores = TemporalFn.forward(self.cx, self.x, a_sub, tm, *params)
        self.d_scores = torch.zeros_like(self.scores)
        self.g_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_bwd, stream=cap, pool=self.g_fwd.pool(), capture_error_mode="thread_local"), torch.no_grad():
            self.outs = TemporalFn.backward(self.cx, self.d_scores)
        torch.cuda.synchronize()

    @staticmethod
    def make_key(tm, feats, a_sub, need_dfeats):
        plist = _temporal_param_list(tm)
        # (the bf16 x 6 launches' grid and K split -- ACX_OPT_X6_CUS / X6_TAIL_SPLIT -- are part of a captured launch: a graph
        # captured under one setting is never replayed under another)
        x6 = ops.x6_options(feats.device.index if feats.device.index is not None else torch.cuda.current_device())
        return (tuple(feats.shape), feats.dtype, None if a_sub is None else a_sub.data_ptr(), bool(need_dfeats),
                tm.precision, int(x6["cus"]), bool(x6["tail_split"])) + tuple(p.data_ptr() for p in plist)


class TemporalGraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, a_sub, tm, *params):
        need = ctx.needs_input_grad[0]
        tg = getattr(tm, "_graphs", None)
        if tg is None or tg.key != _TemporalGraphs.make_key(tm, feats, a_sub, need):
            tg = tm._graphs = _TemporalGraphs(tm, feats, a_sub, need)
        tg.x.copy_(feats)
        tg.g_fwd.replay()
        ctx.tm = tm
        return tg.scores.detach().clone()

    @staticmethod
    def backward(ctx, d_scores):
        tg = ctx.tm._graphs
        tg.d_scores.copy_(d_scores.reshape(tg.d_scores.shape))
        tg.g_bwd.replay()
        # With a gradient sink attached for the duration of this backward (parallel.GradBuckets, armed by the module's
        # train_batch) the graph's static gradient buffers are added into the flat buffer by ONE launch and reported ready
        # there; autograd gets None for them.  Otherwise every gradient is COPIED out of the graph's memory: autograd may keep
        # what a Function returns (torch.autograd.grad, hooks, retain_graph) and the next replay overwrites the static buffers.
        d_feats = tg.outs[0].clone() if tg.outs[0] is not None else None
        sink = getattr(ctx.tm, "_grad_sink", None)
        if sink is not None and sink.accumulate(_temporal_param_list(ctx.tm), tg.outs[3:]):
            return (d_feats, None, None) + (None,) * (len(tg.outs) - 3)
        return (d_feats, None, None) + tuple(g.clone() if g is not None else None for g in tg.outs[3:])


def temporal_train(tm, features, a_sub):
    if getattr(tm, "graph", False) and features.is_cuda:
        return TemporalGraphFn.apply(features.reshape(-1, features.shape[-1]).contiguous(), a_sub, tm, *_temporal_param_list(tm))
    return TemporalFn.apply(features, a_sub, tm, *_temporal_param_list(tm))


class ConcatFeaturesFn(torch.autograd.Function):
    """[logits | x - ncentroid | 0-pad] (anomaly_clip.py:223-233); gradient flows to the logits part only."""

    @staticmethod
    def forward(ctx, logits, x, ncentroid, Kp):
        ctx.C1 = logits.shape[1]
        return ops.concat_features(logits.contiguous(), x, ncentroid, Kp)

    @staticmethod
    def backward(ctx, d_feats):
        return d_feats[:, :ctx.C1].contiguous(), None, None, None


# ====================================================================================================== loss
class MilLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, sim_topk, scores, labels, ia, in_, ba, cfg):
        N, Lg, K, normal_id, lambdas = cfg
        sim, sim_topk, scores = sim.contiguous(), sim_topk.contiguous(), scores.contiguous()
        labels = labels.to(device=sim.device, dtype=torch.int64).contiguous()
        losses, dsim, dtopk, dsc = ops.mil_loss(sim, sim_topk, labels, scores, ia.contiguous(), in_.contiguous(),
                                                ba.contiguous(), N, Lg, K, normal_id, lambdas)
        ctx.cfg = cfg
        ctx.save_for_backward(sim, sim_topk, scores, labels, ia, in_, ba)
        # eight separate outputs (the reference's 8-tuple) with un-materialised gradients: returning ONE [8] tensor and unbinding it
        # outside made autograd build seven zero scalars and stack them for the seven terms nobody differentiates
        ctx.set_materialize_grads(False)
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, d_total, *_unused):
        # Only the total cost (element 0) is meant to be differentiated (module.training_step returns it);
        # the upstream gradient is read on the device, no host sync.
        N, Lg, K, normal_id, lambdas = ctx.cfg
        sim, sim_topk, scores, labels, ia, in_, ba = ctx.saved_tensors
        if d_total is None:
            return (None,) * 8
        gout = d_total.reshape(1).contiguous()
        _, dsim, dtopk, dsc = ops.mil_loss(sim, sim_topk, labels, scores, ia.contiguous(), in_.contiguous(), ba.contiguous(),
                                           N, Lg, K, normal_id, lambdas, gout=gout)
        return dsim, dtopk, dsc, None, None, None, None, None


# ====================================================================================================== assembly
def anomaly_clip_train_forward(net, image_features, labels, ncentroid, masks=None):
    """anomaly_clip.py:156-215."""
    if not net.load_from_features:
        b, t, c, h, w = image_features.size()
        f = net.image_encoder(image_features.view(-1, c, h, w))                    # frozen, forward only
        image_features = f.view(b, net.ncrops, -1, f.shape[-1])
    b, ncrops, t, d = image_features.shape
    if ncrops != 1:
        raise ValueError("training expects ncrops == 1 (the reference squeezes the crop axis, anomaly_clip.py:178-181)")
    x = image_features.reshape(-1, d).contiguous().float()
    sel = net.selector_model
    if masks is None:
        masks = sel.generate_mask(b)
    if x.is_cuda:
        # host-generated masks (selector_model.py:101-117, CPU RNG) go to the device NOW, through pinned memory: a pageable
        # copy issued later would make the host wait for everything queued on the stream before it
        masks = tuple(m if m.is_cuda else m.float().pin_memory().to(x.device, non_blocking=True) for m in masks)
    trainable_text = net.prompt_learner.ctx.requires_grad or net.text_encoder.text_projection.requires_grad
    if getattr(net, "text_graph", False) and trainable_text and x.is_cuda:
        # text tower as two replayed graphs on a side stream (see _TextGraphs); without the logits concat the temporal
        # model does not depend on the selector and runs between the launch and the join
        text_graph_launch(net)
        tf_loc = TextGraphFn.apply(net.prompt_learner.ctx, net.text_encoder.text_projection, net)
        scores = None
        if not net.concat_features:
            feats, a_sub = net.get_temporal_model_input(x, None, ncentroid)
            scores = net.temporal_model(feats, 1, False, a_sub=a_sub).view(-1)
        text_features = text_graph_join(net, tf_loc)
        logits, logits_topk, logits_bottomk, ia, in_, ba = selector_train(sel, x, text_features, labels, ncentroid, masks)
        if scores is None:
            feats, a_sub = net.get_temporal_model_input(x, logits, ncentroid)
            scores = net.temporal_model(feats, 1, False, a_sub=a_sub).view(-1)
        return logits, logits_topk, scores, ia, in_, ba
    text_features = net.get_text_features()
    logits, logits_topk, logits_bottomk, ia, in_, ba = selector_train(sel, x, text_features, labels, ncentroid, masks)
    feats, a_sub = net.get_temporal_model_input(x, logits, ncentroid)
    scores = net.temporal_model(feats, 1, False, a_sub=a_sub).view(-1)
    return logits, logits_topk, scores, ia, in_, ba
