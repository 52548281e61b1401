"""Thin tensor-level wrappers over the libacx C ABI: PyTorch supplies device memory and the
current HIP stream, every arithmetic step runs in the hand-written HIP kernels.  Tensors must be
contiguous CUDA(HIP) tensors; nothing here silently falls back to a torch op."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

_BF16 = torch.bfloat16

# bumped by every in-place parameter update done through raw pointers (AcxAdamW): derived-weight caches
# (bf16 copies, transposes, conv layouts) include it in their keys because such writes do not touch
# torch's tensor version counters.
WEIGHT_EPOCH = [0]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _h(t: torch.Tensor):
    if not t.is_cuda:
        raise L.AcxError("libacx operates on device tensors only (no CPU fallback)")
    return L.ctx(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.ACX_F32
    if t.dtype == _BF16:
        return L.ACX_BF16
    if t.dtype == torch.float16:
        return L.ACX_F16
    raise L.AcxError(f"unsupported dtype {t.dtype}")


SK_MAX_ROWS = 320        # == the library's ACX_OPT_SK_MAX_M default: f32 GEMMs with at most this many rows run the few-row kernel


def set_few_row_limit(device_index: int, rows: int) -> None:
    """ACX_OPT_SK_MAX_M on this device's context and the wrapper's mirror of it (ops.gemm hands such problems no split-K
    workspace, the text-tower glue picks the fused few-row launches below it)."""
    global SK_MAX_ROWS
    h = L.ctx(device_index)
    L.check(L.lib().acx_set_option(h, L.OPT_SK_MAX_M, int(rows)), h)
    SK_MAX_ROWS = int(rows)

def set_x6_cus(device_index: int, cus: int) -> None:
    """ACX_OPT_X6_CUS: cap the persistent bf16 x 6 kernels' grid (0: every CU) so that other streams' kernels find free CUs while
    one of them runs -- they hold a CU's whole register file and 144 KB of its LDS.  Changes the K split (a pure function of
    the shape and of this value): set it BEFORE capturing graphs, and identically on every rank."""
    h = L.ctx(device_index)
    L.check(L.lib().acx_set_option(h, L.OPT_X6_CUS, int(cus)), h)
    x6_options(device_index)["cus"] = int(cus)


X6_MIN_TILES_DEFAULT = 18   # ACX_OPT_X6_MIN_TILES as acx_create sets it (tests restore it)
_X6_OPTS: dict = {}     # device index -> {"cus": n, "tail_split": bool}: the wrapper's mirror of the context options it sizes scratch by


def x6_options(device_index: int) -> dict:
    return _X6_OPTS.setdefault(int(device_index), {"cus": 0, "tail_split": False})


def x6_workgroups(device_index: int) -> int:
    """Workgroups the persistent bf16 x 6 kernels launch on this device: its CU count, or ACX_OPT_X6_CUS when that is lower
    (what acx_gemm's K-split model is evaluated for: the wrapper sizes the split workspace by the same number)."""
    ncu = torch.cuda.get_device_properties(device_index).multi_processor_count
    cap = x6_options(device_index)["cus"]
    return min(ncu, cap) if cap > 0 else ncu


def set_x6_tail_split(device_index: int, on: bool) -> None:
    """ACX_OPT_X6_TAIL_SPLIT: K-split the partly filled last round of tiles of the bf16 x 6 products (default off: it gives the
    tail rows of a launch another summation order than the rows before them)."""
    h = L.ctx(device_index)
    L.check(L.lib().acx_set_option(h, L.OPT_X6_TAIL_SPLIT, int(bool(on))), h)
    x6_options(device_index)["tail_split"] = bool(on)


def set_x6_strip_tail(device_index: int, mode: int) -> None:
    """ACX_OPT_X6_STRIP_TAIL: 1 (default) = a partly filled last round of 256 x 256 tiles is cut into 128- / 64-column strips when
    that shortens it (bit-identical results: same K order per element); 0 = whole tiles only; 2 / 3 = always 128 / 64 columns."""
    h = L.ctx(device_index)
    L.check(L.lib().acx_set_option(h, L.OPT_X6_STRIP_TAIL, int(mode)), h)


def set_x6_min_tiles(device_index: int, tiles: int) -> None:
    """ACX_OPT_X6_MIN_TILES: the ACX_PREC_F32X6 drivers run a product with at least this many 256 x 256 output tiles as a bf16 x 6
    product (smaller ones on the f32 MFMA kernels)."""
    h = L.ctx(device_index)
    L.check(L.lib().acx_set_option(h, L.OPT_X6_MIN_TILES, int(tiles)), h)


_SPLITK_WS: dict = {}
_SPLITK_RETIRED: list = []


def _splitk_workspace(device, nbytes: int) -> torch.Tensor:
    """one reusable split-K scratch buffer per device AND stream (stream-ordered reuse; work on a side stream -- the
    graph-replayed text tower -- must not share scratch with the main stream).  Outgrown buffers stay referenced: a
    captured graph may still hold their address."""
    key = (device.type, device.index, _stream() if device.type == "cuda" else 0)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _SPLITK_RETIRED.append(ws)
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _SPLITK_WS[key] = ws
    return ws


_ZERO_PAGES: dict = {}


def _zero_page(device) -> torch.Tensor:
    """256 bytes of zeros per device, allocated once and never written: the padding source of the conv kernels' LDS-DMA.
    Must not be born inside a stream capture (it would live in that graph's private pool and be re-filled by the captured
    fill kernel only when that graph replays): every capture in this package is preceded by eager warm-up calls, which
    create it; a first use under capture is refused."""
    key = (device.type, device.index)
    z = _ZERO_PAGES.get(key)
    if z is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.AcxError("the conv zero page must be created by an eager call before a HIP graph is captured "
                             "(run the op once outside the capture)")
        z = _ZERO_PAGES[key] = torch.zeros(256, dtype=torch.float32, device=device)     # 1 KB: conv DMA padding + acx_gemm_tn_zp
    return z


_COLSUM_COUNTERS: dict = {}
_CTR_N = 4096            # arrival counters per (device, stream): last-arriver reductions (column sums, split-K pieces, the loss)


def _colsum_counters(device) -> torch.Tensor:
    """4096 uint32 arrival counters per device AND stream for the in-kernel last-arriver reductions (acx_colsum_fused, the
    split-K pieces of acx_gemm, acx_mil_loss_one): zero at rest, every launch resets its own; launches of one stream are
    sequential, so they share the table.  The LAST entry belongs to the loss kernel."""
    key = (device.type, device.index, _stream())
    c = _COLSUM_COUNTERS.get(key)
    if c is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.AcxError("the column-sum counters must be created by an eager call before a HIP graph is captured")
        c = _COLSUM_COUNTERS[key] = torch.zeros(_CTR_N, dtype=torch.int32, device=device)
    return c


def runs_beside(main: "torch.cuda.Stream", cand: "torch.cuda.Stream", device, _big: Optional[torch.Tensor] = None) -> bool:
    """True when work on `cand` overtakes a long kernel on `main`, i.e. the two streams sit on DIFFERENT hardware queues.  The HIP
    streams of a process share GPU_MAX_HW_QUEUES (default 4) hardware queues per priority, a stream takes its queue at first use,
    and a queue runs its packets in order: two streams on one queue serialise.  Measured here (profiles/r06_stream_queues.txt): with
    two or more other streams alive, the step graph's text stream landed on the main stream's queue and configs[1]'s step went from
    10.8 to 15.6 ms.  Test: a 512-MB fill on `main` (~0.1 ms, HBM-bound, few registers: it leaves room on every CU -- a
    register-filling MFMA loop does not, nothing runs beside that on any queue), a 16-byte fill on `cand` that may start once the
    big one has; concurrent streams finish the small fill in ~0.2 of the big one's time, streams on one queue after it.  `cand` is
    used once before the measurement (its first use creates / attaches the queue, ~0.3-6 ms).  The 512-MB buffer lives for the call."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    lib, h = L.lib(), L.ctx(idx)
    big = _big if _big is not None else torch.empty(1 << 27, dtype=torch.float32, device=device)
    tiny = torch.zeros(4, dtype=torch.float32, device=device)
    L.check(lib.acx_fill_f32(h, tiny.data_ptr(), 4, 0.0, cand.cuda_stream), h)
    best = 1.0
    for _ in range(2):
        e0, e1, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(device)
        e0.record(main)
        L.check(lib.acx_fill_f32(h, big.data_ptr(), big.numel(), 0.0, main.cuda_stream), h)
        e1.record(main)
        cand.wait_event(e0)
        L.check(lib.acx_fill_f32(h, tiny.data_ptr(), 4, 1.0, cand.cuda_stream), h)
        ec.record(cand)
        torch.cuda.synchronize(device)
        best = min(best, e0.elapsed_time(ec) / max(e0.elapsed_time(e1), 1e-6))
    return best < 0.4


def side_stream_beside(main: "torch.cuda.Stream", device, priority: int = 0, tries: int = 12) -> "torch.cuda.Stream":
    """A new stream on another hardware queue than `main` (runs_beside): the first of up to `tries` fresh streams that passes the
    test, else the first one created.  For the side streams whose point is concurrency with the caller's stream (the training step's
    text stream, the two-stream ViT's second stream)."""
    if torch.cuda.is_current_stream_capturing():          # the test synchronises: not inside a capture (placement left to chance)
        return torch.cuda.Stream(device=device, priority=priority)
    first = None
    try:
        big = torch.empty(1 << 27, dtype=torch.float32, device=device)
        for _ in range(tries):
            cand = torch.cuda.Stream(device=device, priority=priority)
            if first is None:
                first = cand
            if runs_beside(main, cand, device, big):
                return cand
        return first
    except (RuntimeError, L.AcxError):                     # no room for the test's buffer, ...: a stream all the same
        return first if first is not None else torch.cuda.Stream(device=device, priority=priority)


def prime_capture_stream(stream: "torch.cuda.Stream", device) -> None:
    """Per-stream state that must exist BEFORE a HIP graph is captured on `stream` (zero-initialised buffers cannot be born
    inside a capture): the column-sum arrival counters and the conv zero page."""
    _zero_page(device)
    with torch.cuda.stream(stream):
        _colsum_counters(device)


def cast_bf16(src: torch.Tensor) -> torch.Tensor:
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=_BF16, device=src.device)
    h = _h(src)
    L.check(L.lib().acx_cast_bf16(h, src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), h)
    return dst


def split_bf16x3(src: torch.Tensor, out: Optional[torch.Tensor] = None, panel: bool = False) -> torch.Tensor:
    """f32 [rows, cols] -> bf16 [3, rows, cols]: x = hi + mid + lo to 24 bits (the operands of gemm_x6).  panel: each plane in
    K-panel layout [cols / 32][rows][32] (ACX_BF16X3P; the tensor keeps the shape [3, rows, cols], its memory is panel-ordered)."""
    assert src.dim() == 2 and src.dtype == torch.float32 and src.stride(1) == 1
    rows, cols = src.shape
    if out is None:
        out = torch.empty(3, rows, cols, dtype=_BF16, device=src.device)
    assert out.shape == (3, rows, cols) and out.dtype == _BF16 and out.is_contiguous()
    h = _h(src)
    fn = L.lib().acx_split_bf16x3_panel if panel else L.lib().acx_split_bf16x3
    L.check(fn(h, src.data_ptr(), src.stride(0), out.data_ptr(), rows * cols * 2, rows, cols, _stream()), h)
    return out


_SPLIT_MULTI_ARGS: dict = {}


def split_bf16x3_multi(pairs) -> None:
    """[(src f32 dense, dst bf16 [3, *src.shape]), ...] -> every dst = the three planes of its src, ONE launch (the head's
    convolution weights after an optimizer step).  The pointer tables are host arrays kept alive per pair list."""
    if not pairs:
        return
    key = tuple((s.data_ptr(), d.data_ptr(), s.numel()) for s, d in pairs)
    args = _SPLIT_MULTI_ARGS.get(key)
    if args is None:
        n = len(pairs)
        for s, d in pairs:
            assert s.dtype == torch.float32 and s.is_contiguous() and d.dtype == _BF16 and d.is_contiguous() and d.numel() == 3 * s.numel()
        if len(_SPLIT_MULTI_ARGS) > 64:
            _SPLIT_MULTI_ARGS.clear()
        args = _SPLIT_MULTI_ARGS[key] = (n, (C.c_void_p * n)(*[s.data_ptr() for s, _ in pairs]), (C.c_void_p * n)(*[d.data_ptr() for _, d in pairs]),
                                         (C.c_int64 * n)(*[s.numel() for s, _ in pairs]))
    h = _h(pairs[0][0])
    L.check(L.lib().acx_split_bf16x3_multi(h, args[0], args[1], args[2], args[3], _stream()), h)


def split_f16x2(x: torch.Tensor, panel: bool = False, scale: float = 1.0) -> torch.Tensor:
    """[2, rows, cols] fp16: the two planes hi = fp16(scale x), lo = fp16(scale x - hi) of an f32 matrix (acx_split_f16x2; scale a power
    of two), row-major or K-panel memory order: operands of gemm_x6(..., pairs=3, f16=True) -- the ACX_PREC_F16X3 arithmetic."""
    x = x.contiguous().float()
    rows, cols = x.shape
    out = torch.empty(2, rows, cols, dtype=torch.float16, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_split_f16x2(h, x.data_ptr(), x.stride(0), out.data_ptr(), rows * cols * 2, rows, cols, float(scale), int(panel),
                                    _stream()), h)
    return out


def unpanel(planes: torch.Tensor) -> torch.Tensor:
    """[3, rows, cols] planes in K-panel memory order -> the same values in row-major order (tests / debugging)"""
    npl, rows, cols = planes.shape
    return planes.reshape(npl, cols // 32, rows, 32).permute(0, 2, 1, 3).reshape(npl, rows, cols).contiguous()


def gemm_x6(a3: torch.Tensor, w3: torch.Tensor, *, out: Optional[torch.Tensor] = None, bias=None, act=L.ACT_NONE,
            residual=None, out_dtype=torch.float32, amap=L.AMAP_IDENTITY, gn=0, gl=0, cin=0, M: Optional[int] = None,
            planes_out: bool = False, split_k: bool = True, panels: int = 0, panel_out: bool = False, pairs: int = 6,
            out_scale: float = 0.0) -> torch.Tensor:
    """out[M, N] = epilogue(amap(A) W^T) with A = sum of the three bf16 planes a3 [3, rows, Ka] and W = sum of w3 [3, N, K]
    (split_bf16x3): the six leading cross products on the bf16 matrix cores, f32 accumulation -- the accuracy of an f32
    product (acx_gemm_desc.pairs = 6; acx_gemm_x6.h).  amap = AMAP_CONV3X3: implicit 3x3 convolution over the (gn, gl) token
    grid (K = 9 cin, a3 [3, rows, cin]).  planes_out: the result as three bf16 planes [3, M, N] (the next product's A operand).
    Few output tiles: K is split across workgroups (split_k, workspace owned by this module)."""
    f16 = a3.dtype == torch.float16          # two fp16 planes per operand (split_f16x2): pairs = 3 only, out_scale undoes the planes' scales
    if f16:
        assert pairs == 3 and a3.shape[0] == 2 and w3.shape[0] == 2 and w3.dtype == torch.float16 and (panel_out or not planes_out)
    else:
        assert a3.shape[0] == 3 and w3.shape[0] == 3 and a3.dtype == _BF16 and w3.dtype == _BF16
    assert a3.dim() == 3 and w3.dim() == 3 and a3.is_contiguous() and w3.is_contiguous()
    _, rows, Ka = a3.shape
    N, K = w3.shape[1], w3.shape[2]
    if amap == L.AMAP_CONV3X3:
        assert K == 9 * cin and Ka == cin
    else:
        assert Ka == K
    if M is None:
        M = rows
    if out is None:
        out = torch.empty(((2 if f16 else 3), M, N) if planes_out else (M, N), dtype=(torch.float16 if f16 else _BF16) if planes_out else out_dtype,
                          device=a3.device)
    d = L.GemmDesc()
    d.A, d.W, d.C = a3.data_ptr(), w3.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = Ka, K, out.stride(-2)
    d.a_dtype, d.c_dtype, d.prec = _dt(a3), ((L.F16X2P if f16 else L.BF16X3P if panel_out else L.BF16X3) if planes_out else _dt(out)), L.PREC_BF16
    d.out_scale = float(out_scale)
    d.bias, d.act = _ptr(bias), act
    d.residual, d.ldr = _ptr(residual), (residual.stride(0) if residual is not None else 0)
    d.pairs, d.a_plane_stride, d.w_plane_stride = int(pairs), rows * Ka * 2, N * K * 2      # (pairs = 3: the three leading products only)
    d.panels = panels                      # bit 0: a3 in K-panel memory order (split_bf16x3(panel=True)), bit 1: w3
    d.amap, d.gn, d.gl, d.cin = amap, gn, gl, cin
    if amap == L.AMAP_CONV3X3:
        d.zero_page = _zero_page(a3.device).data_ptr()
    tm, tn = (M + 255) // 256, (N + 255) // 256
    tiles = tm * tn
    dev_i = a3.device.index if a3.device.index is not None else torch.cuda.current_device()
    wgs = x6_workgroups(dev_i)             # (the number acx_gemm evaluates its K-split model for: CUs, or ACX_OPT_X6_CUS)
    ws = None
    if split_k and tiles < wgs and K >= 384:
        ws = _splitk_workspace(a3.device, min(16, max(2, 2 * wgs // tiles)) * M * N * 4)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    elif split_k and x6_options(dev_i)["tail_split"] and amap in (L.AMAP_IDENTITY, L.AMAP_CONV3X3) and K >= 384 and tiles % wgs:
        # ACX_OPT_X6_TAIL_SPLIT only: scratch for the K split of a partly filled LAST round of tiles (acx_gemm splits the launch
        # in two) -- sized from the tail's rows (the tile rows behind the full rounds), up to 4 pieces (8 when the tail is short)
        tail_rows = M - min(M, (tiles // wgs * wgs) // tn * 256)
        if amap == L.AMAP_CONV3X3 and (gn * gl) % 256 == 0:      # (the tail begins at a token-grid boundary)
            gt = gn * gl // 256
            tail_rows = M - min(M, ((tiles // wgs * wgs) // tn) // gt * gt * 256)
        if tail_rows > 0:
            ws = _splitk_workspace(a3.device, min(8, max(4, 2 * wgs // max(1, (tail_rows + 255) // 256 * tn))) * tail_rows * N * 4)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    h = _h(a3)
    L.check(L.lib().acx_gemm(h, C.byref(d), _stream()), h)
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, *, out: Optional[torch.Tensor] = None, M: Optional[int] = None,
         bias=None, act=L.ACT_NONE, residual=None, a_sub=None, prec=L.PREC_F32, out_dtype=torch.float32,
         amap=L.AMAP_IDENTITY, gn=0, gl=0, cin=0, seg=0, pos0=None, pos1=None, a_act=L.ACT_NONE,
         gelu_grad_of=None, tile_table=None, a_norm=None, a_norm_eps=1e-5) -> torch.Tensor:
    """out[M,N] = epilogue(amap(a)[M,K] @ w[N,K]^T); see include/acx.h acx_gemm_desc."""
    assert a.dim() == 2 and w.dim() == 2 and a.is_contiguous() and w.is_contiguous()
    N, K = w.shape
    if M is None:
        M = a.shape[0]
    if amap != L.AMAP_CONV3X3:
        assert a.shape[1] == K, (a.shape, w.shape)
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    d = L.GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = a.stride(0), w.stride(0), out.stride(0)
    d.a_dtype, d.c_dtype, d.prec = _dt(a), _dt(out), prec
    if (prec == L.PREC_BF16) != (w.dtype == _BF16):
        raise L.AcxError("weight dtype does not match the requested MFMA precision")
    d.bias, d.act = _ptr(bias), act
    d.residual, d.ldr = _ptr(residual), (residual.stride(0) if residual is not None else 0)
    d.a_sub = _ptr(a_sub)
    d.amap, d.gn, d.gl, d.cin, d.seg = amap, gn, gl, cin, seg
    d.pos0, d.pos1 = _ptr(pos0), _ptr(pos1)
    d.a_act = a_act
    if amap == L.AMAP_CONV3X3:
        d.zero_page = _zero_page(a.device).data_ptr()
    d.gelu_grad_of, d.ldg = _ptr(gelu_grad_of), (gelu_grad_of.stride(0) if gelu_grad_of is not None else 0)
    if a_norm is not None:                                   # (weight, bias): LayerNorm over K in the few-row kernel's A prologue
        d.a_norm_w, d.a_norm_b, d.a_norm_eps = a_norm[0].data_ptr(), a_norm[1].data_ptr(), float(a_norm_eps)
    if amap == L.AMAP_TILETABLE:
        assert tile_table is not None and tile_table.dtype == torch.int32 and tile_table.is_cuda and tile_table.is_contiguous()
        assert tile_table.numel() == 2 * (M // (gn * gl))
        d.tile_table = tile_table.data_ptr()
    ws = None
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    few_rows = ((M <= SK_MAX_ROWS or (N <= 512 and M <= 4 * SK_MAX_ROWS and not (M > 768 and K >= 1536)))
                and K % 256 == 0 and prec == L.PREC_F32
                and amap == L.AMAP_IDENTITY and a_sub is None and pos0 is None
                and act != L.ACT_LEAKYRELU)                                   # acx_gemm takes the few-row kernel: no split-K
    # split-K candidates: <= 128 output tiles, or up to 256 with a long K (the convolutions of a data-parallel rank's 4096
    # rows: 256 tiles = ONE 8-wave block per CU; two K halves put two blocks on every CU for the price of a 12 us reduce)
    generic_f32 = (a_sub is not None or pos0 is not None or amap in (L.AMAP_TESTTILE, L.AMAP_TILETABLE)) and prec == L.PREC_F32 \
        and a.dtype == torch.float32 and out.dtype == torch.float32
    if ((tiles <= 128 and K >= 256 or tiles <= 256 and K >= 1024) and not few_rows
            and ((amap in (L.AMAP_IDENTITY, L.AMAP_CONV3X3) and a_sub is None and pos0 is None) or generic_f32)):
        ws = _splitk_workspace(a.device, min(16, 512 // tiles) * M * N * 4)     # skinny problem: let the library split K
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        if amap == L.AMAP_CONV3X3 and prec == L.PREC_F32:
            # f32 convolutions: the split pieces are reduced INSIDE the kernel (last-arriver, no second launch)
            ctr = _colsum_counters(a.device)
            d.counters, d.n_counters = ctr.data_ptr(), _CTR_N - 1
    if few_rows and K >= 1024 and K % 512 == 0 and ((M + 31) // 32) * ((N + 31) // 32) * (K // 512) <= 512:
        # long K on few tiles (the text tower's K = 2048 GEMMs): K split across workgroups, last-arriver reduction in-kernel
        tiles32 = ((M + 31) // 32) * ((N + 31) // 32)
        ws = _splitk_workspace(a.device, (K // 512) * tiles32 * 4096)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        ctr = _colsum_counters(a.device)
        d.counters, d.n_counters = ctr.data_ptr(), _CTR_N - 1               # the last entry belongs to the loss kernel
    h = _h(a)
    L.check(L.lib().acx_gemm(h, C.byref(d), _stream()), h)
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *, eps=1e-5, mode=L.NORM_LAYER,
              out_dtype=torch.float32, rows: Optional[int] = None, ldx: Optional[int] = None, planes_out: bool = False,
              panel_out: bool = False) -> torch.Tensor:
    """planes_out: the result as three bf16 planes [3, rows, D] (hi | mid | lo of the f32 value: a bf16 x 6 product's operand);
    panel_out: each plane in K-panel memory order (ACX_BF16X3P; unpanel() for row-major)"""
    D = w.numel()
    if rows is None:
        rows = x.numel() // D
        ldx = D
    y = torch.empty((3, rows, D) if planes_out else (rows, D), dtype=_BF16 if planes_out else out_dtype, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_layernorm(h, x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), y.data_ptr(), D,
                                  (L.BF16X3P if panel_out else L.BF16X3) if planes_out else _dt(y), rows, D, eps, mode, _stream()), h)
    return y


def attention(qkv: torch.Tensor, batch: int, L_: int, heads: int, causal: bool) -> torch.Tensor:
    assert qkv.is_contiguous() and qkv.shape == (batch * L_, 3 * heads * 64)
    out = torch.empty(batch * L_, heads * 64, dtype=torch.float32, device=qkv.device)
    h = _h(qkv)
    L.check(L.lib().acx_attention(h, qkv.data_ptr(), qkv.stride(0), out.data_ptr(), out.stride(0), batch, L_, heads,
                                  int(causal), _stream()), h)
    return out


def attention_p3(qkv3: torch.Tensor, batch: int, L_: int, heads: int, products: int = 6) -> torch.Tensor:
    """the ViT attention on the bf16 matrix cores at f32 accuracy (acx_attention_p3): qkv3 [3, batch * L, 3 * heads * 64] = the
    three bf16 planes of q | k | v in K-panel memory order (split_bf16x3(panel=True)); returns the three planes of the output
    [3, batch * L, heads * 64], K-panel memory order as well (unpanel() for row-major)."""
    W = heads * 64
    if products == 103:                                   # two fp16 planes (split_f16x2(panel=True)), three products: the "f16x3" arithmetic
        assert qkv3.dtype == torch.float16 and qkv3.is_contiguous() and qkv3.shape == (2, batch * L_, 3 * W)
        out = torch.empty(2, batch * L_, W, dtype=torch.float16, device=qkv3.device)
        h = _h(qkv3)
        L.check(L.lib().acx_attention_p3n(h, qkv3.data_ptr(), out.data_ptr(), batch, L_, heads, 103, _stream()), h)
        return out
    assert qkv3.dtype == _BF16 and qkv3.is_contiguous() and qkv3.shape == (3, batch * L_, 3 * W)
    out = (torch.zeros if products == 3 else torch.empty)(3, batch * L_, W, dtype=_BF16, device=qkv3.device)   # (3 products: lo plane unwritten)
    h = _h(qkv3)
    L.check(L.lib().acx_attention_p3n(h, qkv3.data_ptr(), out.data_ptr(), batch, L_, heads, int(products), _stream()), h)
    return out


def attention_bf16(qkv: torch.Tensor, batch: int, L_: int, heads: int) -> torch.Tensor:
    """bf16 attention (bf16 mode only): qkv [batch*L, 3*heads*64] bf16 -> [batch*L, heads*64] bf16."""
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    W = heads * 64
    out = torch.empty(batch * L_, W, dtype=torch.bfloat16, device=qkv.device)
    h = _h(qkv)
    L.check(L.lib().acx_attention_bf16(h, qkv.data_ptr(), qkv.stride(0), out.data_ptr(), W, batch, L_, heads, _stream()), h)
    return out


def text_directions(text: torch.Tensor, ncentroid: torch.Tensor, normal_id: int) -> torch.Tensor:
    Cc, D = text.shape
    dirs = torch.empty(Cc - 1, D, dtype=torch.float32, device=text.device)
    h = _h(text)
    L.check(L.lib().acx_text_directions(h, text.data_ptr(), ncentroid.data_ptr(), dirs.data_ptr(), Cc, D, normal_id,
                                        _stream()), h)
    return dirs


def selector_project(x: torch.Tensor, ncentroid: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    x = x.reshape(-1, x.shape[-1])
    assert x.is_contiguous()
    rows, D = x.shape
    C1 = dirs.shape[0]
    raw = torch.empty(rows, C1, dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_selector_project(h, x.data_ptr(), ncentroid.data_ptr(), dirs.data_ptr(), raw.data_ptr(), rows, D,
                                         C1, _stream()), h)
    return raw


def bn_combine(gathered: torch.Tensor, C1: int):
    """SyncBN combine: gathered [R, 2 C1 + 1] (per rank: mean, biased var * rows, rows) -> (mean, var_biased, var_unbiased,
    total rows as a device scalar [1]) in one launch."""
    assert gathered.is_cuda and gathered.dtype == torch.float32 and gathered.is_contiguous() and gathered.shape[1] == 2 * C1 + 1
    out = torch.empty(3 * C1 + 1, dtype=torch.float32, device=gathered.device)
    h = _h(gathered)
    L.check(L.lib().acx_bn_combine(h, gathered.data_ptr(), gathered.shape[0], C1, out.data_ptr(), out[C1:].data_ptr(),
                                   out[2 * C1:].data_ptr(), out[3 * C1:].data_ptr(), _stream()), h)
    return out[:C1], out[C1:2 * C1], out[2 * C1:3 * C1], out[3 * C1:]


def _bn_workspace(t: torch.Tensor, rows: int, C1: int) -> torch.Tensor:
    return torch.empty(int(L.lib().acx_bn_workspace_bytes(rows, C1)) // 8, dtype=torch.float64, device=t.device)


def bn_stats(raw: torch.Tensor):
    rows, C1 = raw.shape
    st = torch.empty(3, C1, dtype=torch.float32, device=raw.device)
    h = _h(raw)
    ws = _bn_workspace(raw, rows, C1)
    L.check(L.lib().acx_bn_stats(h, raw.data_ptr(), rows, C1, st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
                                 ws.data_ptr(), ws.numel() * 8, _stream()), h)
    return st[0], st[1], st[2]


def selector_project_stats(x: torch.Tensor, ncentroid: torch.Tensor, dirs: torch.Tensor):
    """projection + batch statistics of its output in the projection's epilogue -> (raw, mean, var_biased, var_unbiased)."""
    x = x.reshape(-1, x.shape[-1])
    assert x.is_contiguous()
    rows, D = x.shape
    C1 = dirs.shape[0]
    raw = torch.empty(rows, C1, dtype=torch.float32, device=x.device)
    st = torch.empty(3, C1, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x, rows, C1)
    h = _h(x)
    L.check(L.lib().acx_selector_project_stats(h, x.data_ptr(), ncentroid.data_ptr(), dirs.data_ptr(), raw.data_ptr(), rows, D,
                                               C1, st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), ws.data_ptr(),
                                               ws.numel() * 8, _stream()), h)
    return raw, st[0], st[1], st[2]


def selector_bn(raw: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, eps=1e-5,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows, C1 = raw.shape
    if out is None:
        out = torch.empty(rows, C1, dtype=torch.float32, device=raw.device)
    h = _h(raw)
    L.check(L.lib().acx_selector_bn(h, raw.data_ptr(), mean.data_ptr(), var.data_ptr(), out.data_ptr(), out.stride(0),
                                    rows, C1, eps, _stream()), h)
    return out


def axial_attention(qkv: torch.Tensor, tiles: int, gn: int, gl: int, heads: int, e: int, axis: int) -> torch.Tensor:
    assert qkv.is_contiguous() and qkv.shape == (tiles * gn * gl, 3 * heads * e)
    out = torch.empty(tiles * gn * gl, heads * e, dtype=torch.float32, device=qkv.device)
    h = _h(qkv)
    L.check(L.lib().acx_axial_attention(h, qkv.data_ptr(), out.data_ptr(), tiles, gn, gl, heads, e, axis, _stream()), h)
    return out


def cls_head(x1, x2, ln_w, ln_b, lin_w, lin_b, gn: int, gl: int, seg: int, tile_table=None) -> torch.Tensor:
    rows, E = x1.shape
    scores = torch.empty(rows, dtype=torch.float32, device=x1.device)
    h = _h(x1)
    if tile_table is not None:
        assert tile_table.dtype == torch.int32 and tile_table.is_cuda and tile_table.numel() == 2 * (rows // (gn * gl))
        L.check(L.lib().acx_cls_head_tiles(h, x1.data_ptr(), x2.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), lin_w.data_ptr(),
                                           lin_b.data_ptr(), scores.data_ptr(), rows, E, gn, gl, tile_table.data_ptr(), _stream()), h)
        return scores
    L.check(L.lib().acx_cls_head(h, x1.data_ptr(), x2.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), lin_w.data_ptr(),
                                 lin_b.data_ptr(), scores.data_ptr(), rows, E, gn, gl, seg, _stream()), h)
    return scores


def class_probs(sim: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
    rows, C1 = sim.shape
    probs = torch.empty_like(sim)
    h = _h(sim)
    L.check(L.lib().acx_class_probs(h, sim.data_ptr(), scores.data_ptr(), probs.data_ptr(), rows, C1, _stream()), h)
    return probs


def colsum_(acc: torch.Tensor, x: torch.Tensor) -> None:
    """acc += column sums of x (the ncentroid accumulation, anomaly_clip_module.py:145-171): the deterministic two-stage
    column sum (row-slab partials, fixed-order reduce) + one axpby.  (acx_colsum -- one float atomicAdd per column per
    256-row slab -- serialises 65 k atomics on 512 addresses at the benchmark shape: 78 us for 67 MB, order-dependent bits.)"""
    x = x.reshape(-1, x.shape[-1])
    assert x.is_contiguous()
    colsum(x, acc=acc)


def prompt_embed(prefix, ctxv, suffix, pos: Optional[torch.Tensor], n_ctx: int, Lout: Optional[int] = None) -> torch.Tensor:
    """[C, Lout, W] prompts (+ positional embedding); Lout < Lc keeps the first Lout positions only."""
    Cc, _, W = prefix.shape
    Lc = 1 + n_ctx + suffix.shape[1]
    Lout = Lc if Lout is None else min(int(Lout), Lc)
    assert prefix.is_contiguous() and suffix.is_contiguous() and ctxv.is_contiguous()
    out = torch.empty(Cc, Lout, W, dtype=torch.float32, device=prefix.device)
    h = _h(prefix)
    L.check(L.lib().acx_prompt_embed(h, prefix.data_ptr(), ctxv.data_ptr(), suffix.data_ptr(), _ptr(pos), out.data_ptr(),
                                     Cc, n_ctx, Lc, W, int(ctxv.dim() == 2), Lout, _stream()), h)
    return out


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    assert x.dim() == 2 and x.is_contiguous() and idx.dtype == torch.int64 and idx.is_cuda
    out = torch.empty(idx.numel(), x.shape[1], dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_gather_rows(h, x.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), x.shape[1], _stream()), h)
    return out


def add_bcast(x: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    assert x.is_contiguous() and p.is_contiguous()
    out = torch.empty_like(x)
    h = _h(x)
    L.check(L.lib().acx_add_bcast(h, x.data_ptr(), p.data_ptr(), out.data_ptr(), x.numel() // p.numel(), p.numel(),
                                  _stream()), h)
    return out


def concat_features(logits: torch.Tensor, x: torch.Tensor, ncentroid: torch.Tensor, Kp: int) -> torch.Tensor:
    rows, C1 = logits.shape
    D = x.shape[-1]
    out = torch.empty(rows, Kp, dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_concat_features(h, logits.data_ptr(), x.data_ptr(), ncentroid.data_ptr(), out.data_ptr(), rows, C1,
                                        D, Kp, _stream()), h)
    return out


# ---------------------------------------------------------------------------------------------------
# training-side wrappers
def gemm_tn(a: torch.Tensor, b: torch.Tensor, *, b_sub=None, conv=False, gn=0, gl=0, cin=0, N2: Optional[int] = None,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[N1,N2] = sum_m a[m,n1] * bmap(b)[m,n2]  (weight gradient dW = dY^T X).  `out`: a dense [N1, N2] f32 destination (a
    gradient view of parallel.GradBuckets.flat: the weight gradient is produced in place)."""
    assert a.dim() == 2 and b.dim() == 2 and a.is_contiguous() and b.is_contiguous() and a.shape[0] == b.shape[0]
    M, N1 = a.shape
    if N2 is None:
        N2 = 9 * cin if conv else b.shape[1]
    if out is None:
        out = torch.empty(N1, N2, dtype=torch.float32, device=a.device)
    else:
        assert out.shape == (N1, N2) and out.is_contiguous() and out.dtype == torch.float32
    lib = L.lib()
    nbytes = lib.acx_gemm_tn_workspace_bytes(M, N1, N2)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=a.device)
    h = _h(a)
    L.check(lib.acx_gemm_tn_zp(h, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N2, M, N1, N2, _ptr(b_sub),
                               int(conv), gn, gl, cin, ws.data_ptr(), ws.numel(), _zero_page(a.device).data_ptr(), _stream()), h)
    return out


def gemm_tn_x6(a3: torch.Tensor, b3: torch.Tensor, *, conv=False, gn=0, gl=0, cin=0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[N1, N2] = sum_m A[m, n1] * bmap(B)[m, n2] with A = sum of the planes a3 [3, M, N1], B = sum of b3 [3, M, cin or N2]
    (split_bf16x3): the weight gradient as an f32-accurate product on the bf16 matrix cores (acx_gemm_tn_x6).  N1 a multiple of
    256 and N2 of 128, or N1 of 128 and N2 of 256 (conv: cin a multiple of the tile width the library picks: 256 / 128)."""
    assert a3.dim() == 3 and b3.dim() == 3 and a3.shape[0] == 3 and b3.shape[0] == 3 and a3.dtype == _BF16 and b3.dtype == _BF16
    assert a3.is_contiguous() and b3.is_contiguous() and a3.shape[1] == b3.shape[1]
    _, M, N1 = a3.shape
    N2 = 9 * cin if conv else b3.shape[2]
    if out is None:
        out = torch.empty(N1, N2, dtype=torch.float32, device=a3.device)
    else:
        assert out.shape == (N1, N2) and out.is_contiguous() and out.dtype == torch.float32
    lib = L.lib()
    ws = _splitk_workspace(a3.device, int(lib.acx_gemm_tn_x6_workspace_bytes(M, N1, N2)))
    h = _h(a3)
    L.check(lib.acx_gemm_tn_x6(h, a3.data_ptr(), M * N1 * 2, N1, b3.data_ptr(), M * b3.shape[2] * 2, b3.shape[2], out.data_ptr(), N2,
                               M, N1, N2, int(conv), gn, gl, cin, ws.data_ptr(), ws.numel(), _zero_page(a3.device).data_ptr(),
                               _stream()), h)
    return out


def gemm_tn_group(problems):
    """problems: list of (a [M, N1], b [M, N2], out or None, b_sub or None) -> list of C_k = a_k^T (b_k - b_sub_k); ONE launch
    (+ one reduce) for all of them, bit-identical to separate gemm_tn calls (acx_gemm_tn_group)."""
    n = len(problems)
    if n == 0:
        return []
    arr = (L.TnProblem * n)()
    outs = []
    for i, (a, b, out, b_sub) in enumerate(problems):
        assert a.dim() == 2 and b.dim() == 2 and a.is_contiguous() and b.is_contiguous() and a.shape[0] == b.shape[0]
        M, N1 = a.shape
        N2 = b.shape[1]
        if out is None:
            out = torch.empty(N1, N2, dtype=torch.float32, device=a.device)
        else:
            assert out.shape == (N1, N2) and out.is_contiguous() and out.dtype == torch.float32
        outs.append(out)
        arr[i].A, arr[i].B, arr[i].C, arr[i].b_sub = a.data_ptr(), b.data_ptr(), out.data_ptr(), _ptr(b_sub)
        arr[i].M, arr[i].N1, arr[i].N2, arr[i].lda, arr[i].ldb = M, N1, N2, a.stride(0), b.stride(0)
    lib = L.lib()
    pa = C.cast(arr, C.c_void_p)
    nbytes = int(lib.acx_gemm_tn_group_workspace_bytes(n, pa))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=problems[0][0].device)
    h = _h(problems[0][0])
    L.check(lib.acx_gemm_tn_group(h, n, pa, ws.data_ptr(), ws.numel(), _stream()), h)
    return outs


def colsum_group(xs):
    """column sums of several [rows_i, D_i] matrices in ONE launch (acx_colsum_fused_group) -> list of [D_i] tensors;
    bit-identical to colsum() of each."""
    n = len(xs)
    if n == 0:
        return []
    lib = L.lib()
    dev = xs[0].device
    outs, parts = [], []
    for x in xs:
        assert x.dim() == 2 and x.is_contiguous() and x.shape[1] % 4 == 0 and x.data_ptr() % 16 == 0
        rows, D = x.shape
        outs.append(torch.empty(D, dtype=torch.float32, device=dev))
        parts.append(torch.empty(max(int(lib.acx_colsum_fused_part_bytes(rows, D)) // 4, 4), dtype=torch.float32, device=dev))
    ctr = _colsum_counters(dev)
    h = _h(xs[0])
    vp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])            # noqa: E731
    L.check(lib.acx_colsum_fused_group(h, n, vp(xs), (C.c_int32 * n)(*[x.stride(0) for x in xs]),
                                       (C.c_int64 * n)(*[x.shape[0] for x in xs]), (C.c_int32 * n)(*[x.shape[1] for x in xs]),
                                       vp(outs), vp(parts), ctr.data_ptr(), ctr.numel() - 1, _stream()), h)
    return outs


def reduce_rows_group(parts):
    """reduce_rows of several partial tables in ONE launch -> list of [width_i] tensors (same summation tree)."""
    n = len(parts)
    if n == 0:
        return []
    outs = [torch.empty(p.shape[1], dtype=torch.float32, device=p.device) for p in parts]
    h = _h(parts[0])
    vp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])            # noqa: E731
    L.check(L.lib().acx_reduce_rows_group(h, n, vp(parts), vp(outs), (C.c_int32 * n)(*[p.shape[0] for p in parts]),
                                          (C.c_int32 * n)(*[p.shape[1] for p in parts]), _stream()), h)
    return outs


def reduce_rows(part: torch.Tensor) -> torch.Tensor:
    nparts, width = part.shape
    out = torch.empty(width, dtype=torch.float32, device=part.device)
    h = _h(part)
    L.check(L.lib().acx_reduce_rows(h, part.data_ptr(), out.data_ptr(), nparts, width, _stream()), h)
    return out


def layernorm_bwd(x, w, dy, *, eps=1e-5, mode=L.NORM_LAYER, need_dx=True, need_params=True, dx_scale=1.0, add=None):
    """returns (dx or None, dw or None, db or None); `add` [rows, D]: dx = add + dx_scale * dL/dx in the same pass."""
    D = w.numel()
    x = x.reshape(-1, D)
    dy = dy.reshape(-1, D)
    assert x.is_contiguous() and dy.is_contiguous()
    rows = x.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    part = torch.empty(int(L.lib().acx_row_parts(rows)), 2 * D, dtype=torch.float32, device=x.device) if need_params else None
    h = _h(x)
    if add is not None:
        add = add.reshape(-1, D)
        assert need_dx and add.is_contiguous() and add.shape == x.shape
    L.check(L.lib().acx_layernorm_bwd(h, x.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dx), _ptr(part), rows, D, eps, mode,
                                      dx_scale, _ptr(add), _stream()), h)
    if need_params:
        s = reduce_rows(part)
        return dx, s[:D], s[D:]
    return dx, None, None


def layernorm_bwd_parts(x, w, dy, *, eps=1e-5, mode=L.NORM_LAYER, dx_scale=1.0, add=None):
    """layernorm_bwd with the parameter-gradient reduction left to the caller: returns (dx, part [acx_row_parts(rows), 2 D]);
    reduce_rows(part) = [dw | db].  The step graph runs that reduction on its weight-gradient side branch."""
    D = w.numel()
    x = x.reshape(-1, D)
    dy = dy.reshape(-1, D)
    assert x.is_contiguous() and dy.is_contiguous()
    rows = x.shape[0]
    dx = torch.empty_like(x)
    part = torch.empty(int(L.lib().acx_row_parts(rows)), 2 * D, dtype=torch.float32, device=x.device)
    h = _h(x)
    if add is not None:
        add = add.reshape(-1, D)
        assert add.is_contiguous() and add.shape == x.shape
    L.check(L.lib().acx_layernorm_bwd(h, x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), part.data_ptr(), rows, D, eps, mode,
                                      dx_scale, _ptr(add), _stream()), h)
    return dx, part


def cls_head_bwd_parts(x1, x2, ln_w, ln_b, lin_w, scores, dscores):
    """cls_head_bwd without the final reduction: (dx, part [acx_row_parts(rows), 3 E + 4])."""
    rows, E = x1.shape
    dx = torch.empty_like(x1)
    part = torch.empty(int(L.lib().acx_row_parts(rows)), 3 * E + 4, dtype=torch.float32, device=x1.device)
    h = _h(x1)
    L.check(L.lib().acx_cls_head_bwd(h, x1.data_ptr(), x2.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), lin_w.data_ptr(),
                                     scores.data_ptr(), dscores.data_ptr(), dx.data_ptr(), part.data_ptr(), rows, E, _stream()), h)
    return dx, part


def cls_head_bwd(x1, x2, ln_w, ln_b, lin_w, scores, dscores):
    rows, E = x1.shape
    dx = torch.empty_like(x1)
    PW = 3 * E + 4
    part = torch.empty(int(L.lib().acx_row_parts(rows)), PW, dtype=torch.float32, device=x1.device)
    h = _h(x1)
    L.check(L.lib().acx_cls_head_bwd(h, x1.data_ptr(), x2.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), lin_w.data_ptr(),
                                     scores.data_ptr(), dscores.data_ptr(), dx.data_ptr(), part.data_ptr(), rows, E, _stream()), h)
    s = reduce_rows(part)
    return dx, s[:E], s[E:2 * E], s[2 * E:3 * E], s[3 * E:3 * E + 1]


def act(saved: torch.Tensor, d: Optional[torch.Tensor], mode: int) -> torch.Tensor:
    out = torch.empty_like(saved)
    h = _h(saved)
    L.check(L.lib().acx_act(h, saved.data_ptr(), _ptr(d), out.data_ptr(), saved.numel(), mode, _stream()), h)
    return out


def leaky_grad_planes(u3: torch.Tensor, d: torch.Tensor):
    """LeakyReLU backward from the three-plane activation u3 [3, rows, C] (only its hi plane is read): returns
    (d_pre [rows, C] f32, d_pre3 [3, rows, C] bf16 planes)."""
    assert u3.dim() == 3 and u3.shape[0] == 3 and u3.dtype == _BF16 and u3.is_contiguous() and d.is_contiguous() and d.shape == u3.shape[1:]
    out = torch.empty_like(d)
    planes = torch.empty_like(u3)
    h = _h(d)
    L.check(L.lib().acx_leaky_grad_planes(h, u3.data_ptr(), d.data_ptr(), out.data_ptr(), planes.data_ptr(), d.numel(), d.numel(), _stream()), h)
    return out, planes


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    out = torch.empty_like(a)
    h = _h(a)
    L.check(L.lib().acx_add(h, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), h)
    return out


def transpose(x: torch.Tensor) -> torch.Tensor:
    assert x.dim() == 2 and x.is_contiguous()
    out = torch.empty(x.shape[1], x.shape[0], dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_transpose(h, x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _stream()), h)
    return out


def conv_weight_dx(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [Cin, 9*Cout] (flipped taps) for the dX implicit GEMM."""
    Cout, Cin = w.shape[0], w.shape[1]
    out = torch.empty(Cin, 9 * Cout, dtype=torch.float32, device=w.device)
    h = _h(w)
    L.check(L.lib().acx_conv_weight_dx(h, w.data_ptr(), out.data_ptr(), Cout, Cin, _stream()), h)
    return out


def seq_attention_bwd(qkv, dout, tiles, gn, gl, heads, e, axis, causal=False) -> torch.Tensor:
    dqkv = torch.empty_like(qkv)
    h = _h(qkv)
    stats = torch.empty(qkv.shape[0] * heads * 3, dtype=torch.float32, device=qkv.device)
    L.check(L.lib().acx_seq_attention_bwd(h, qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), tiles, gn, gl, heads, e, axis,
                                          int(causal), stats.data_ptr(), _stream()), h)
    return dqkv


def pos_grad(dx, tiles, gn, gl):
    E = dx.shape[1]
    d0 = torch.empty(gn, E, dtype=torch.float32, device=dx.device)
    d1 = torch.empty(gl, E, dtype=torch.float32, device=dx.device)
    part = torch.empty(tiles * gn * E + tiles * ((gn + 3) // 4) * gl * E, dtype=torch.float32, device=dx.device)
    h = _h(dx)
    L.check(L.lib().acx_pos_grad(h, dx.data_ptr(), d0.data_ptr(), d1.data_ptr(), part.data_ptr(), tiles, gn, gl, E, _stream()), h)
    return d0, d1


def bn_bwd_stats(logits, dlogits) -> torch.Tensor:
    rows, C1 = logits.shape
    sums = torch.empty(2 * C1, dtype=torch.float32, device=logits.device)
    h = _h(logits)
    ws = _bn_workspace(logits, rows, C1)
    L.check(L.lib().acx_bn_bwd_stats(h, logits.data_ptr(), dlogits.data_ptr(), sums.data_ptr(), rows, C1, ws.data_ptr(),
                                     ws.numel() * 8, _stream()), h)
    return sums


def bn_bwd_apply(logits, dlogits, var_biased, sums, total_rows, eps=1e-5, pad_to: int = 4) -> torch.Tensor:
    """returns draw padded to a multiple of `pad_to` columns (zero pad) so it can feed acx_gemm_tn.  `total_rows`: a host
    int, or a device f32 scalar tensor (SyncBN: the all-gathered row count never visits the host)."""
    rows, C1 = logits.shape
    C1p = (C1 + pad_to - 1) // pad_to * pad_to
    draw = torch.empty(rows, C1p, dtype=torch.float32, device=logits.device)      # (the kernel zeroes the pad columns)
    h = _h(logits)
    dev_n = total_rows if torch.is_tensor(total_rows) else None
    if dev_n is not None:
        assert dev_n.is_cuda and dev_n.dtype == torch.float32 and dev_n.numel() == 1
    L.check(L.lib().acx_bn_bwd_apply(h, logits.data_ptr(), dlogits.data_ptr(), var_biased.data_ptr(), sums.data_ptr(),
                                     draw.data_ptr(), C1p, rows, 0 if dev_n is not None else int(total_rows), C1, eps,
                                     _ptr(dev_n), _stream()), h)
    return draw


def row_parts(rows: int) -> int:
    return int(L.lib().acx_row_parts(int(rows)))


def fill_(t: torch.Tensor, value: float = 0.0) -> torch.Tensor:
    """t[:] = value through a libacx kernel (capturable; hipMemsetAsync nodes and torch fill kernels stay out of the step graphs)."""
    assert t.is_contiguous() and t.dtype == torch.float32
    h = _h(t)
    L.check(L.lib().acx_fill_f32(h, t.data_ptr(), t.numel(), float(value), _stream()), h)
    return t


def zeros(*shape, device) -> torch.Tensor:
    return fill_(torch.empty(*shape, dtype=torch.float32, device=device), 0.0)


def prep_multi(segs, device=None) -> None:
    """segs: iterable of (src, dst, rows, cols, src_ld, dst_ld, transpose) with src / dst f32 device tensors (or data pointers):
    every strided copy / transpose in ONE launch (acx_prep_multi)."""
    segs = list(segs)
    if not segs:
        return
    arr = (L.PrepSeg * len(segs))()
    dev_t = None
    for i, (src, dst, rows, cols, sld, dld, tr) in enumerate(segs):
        if torch.is_tensor(src):
            dev_t = src
        arr[i].src = src.data_ptr() if torch.is_tensor(src) else int(src)
        arr[i].dst = dst.data_ptr() if torch.is_tensor(dst) else int(dst)
        arr[i].rows, arr[i].cols, arr[i].src_ld, arr[i].dst_ld, arr[i].transpose = int(rows), int(cols), int(sld), int(dld), int(bool(tr))
    if device is not None:
        if device.type != "cuda":
            raise L.AcxError("libacx operates on device tensors only (no CPU fallback)")
        h = L.ctx(device.index if device.index is not None else torch.cuda.current_device())
    else:
        h = _h(dev_t)
    L.check(L.lib().acx_prep_multi(h, len(segs), C.cast(arr, C.c_void_p), _stream()), h)


def multi_copy_(ys, xs) -> None:
    """y_i = x_i for lists of equally sized contiguous f32 tensors, one launch."""
    n = len(ys)
    if n == 0:
        return
    assert all(y.numel() == x.numel() and y.is_contiguous() and x.is_contiguous() for y, x in zip(ys, xs))
    h = _h(ys[0])
    L.check(L.lib().acx_multi_copy(h, n, (C.c_void_p * n)(*[y.data_ptr() for y in ys]), (C.c_void_p * n)(*[x.data_ptr() for x in xs]),
                                   (C.c_int64 * n)(*[y.numel() for y in ys]), _stream()), h)


def adamw_hyper(lrs, wds, beta1, beta2, step: int, out: torch.Tensor) -> None:
    """fills `out` (CPU float32 [1 + 2 n], typically pinned) with the scalars of one AdamW step (acx_adamw_hyper)."""
    n = len(lrs)
    assert out.dtype == torch.float32 and out.numel() >= 1 + 2 * n and not out.is_cuda and out.is_contiguous()
    rc = L.lib().acx_adamw_hyper(n, (C.c_double * n)(*[float(x) for x in lrs]), (C.c_double * n)(*[float(x) for x in wds]),
                                 float(beta1), float(beta2), int(step), C.cast(out.data_ptr(), C.POINTER(C.c_float)))
    if rc != 0:
        raise L.AcxError(f"acx_adamw_hyper failed [{rc}]")


def adamw_multi_dev_(ps, gs, ms, vs, hyper_dev: torch.Tensor, grad_scale: float, beta1, beta2, eps) -> None:
    """acx_adamw_multi with per-tensor scalars in device memory (`hyper_dev` f32 [1 + 2 n], see adamw_hyper)."""
    n = len(ps)
    if n == 0:
        return
    assert hyper_dev.is_cuda and hyper_dev.dtype == torch.float32 and hyper_dev.numel() >= 1 + 2 * n
    h = _h(ps[0])
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])           # noqa: E731
    L.check(L.lib().acx_adamw_multi_dev(h, n, arr(ps), arr(gs), arr(ms), arr(vs), (C.c_int64 * n)(*[p.numel() for p in ps]),
                                        hyper_dev.data_ptr(), float(grad_scale), beta1, beta2, eps, _stream()), h)


def bn_pack(mean: torch.Tensor, var_b: torch.Tensor, rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    C1 = mean.numel()
    if out is None:
        out = torch.empty(2 * C1 + 1, dtype=torch.float32, device=mean.device)
    h = _h(mean)
    L.check(L.lib().acx_bn_pack(h, mean.data_ptr(), var_b.data_ptr(), int(rows), C1, out.data_ptr(), _stream()), h)
    return out


def bn_running_update_(bn, mean: torch.Tensor, var_u: torch.Tensor) -> None:
    """nn.BatchNorm1d's running_mean / running_var / num_batches_tracked update in one launch."""
    h = _h(mean)
    nbt = bn.num_batches_tracked
    assert nbt.dtype == torch.int64 and nbt.is_cuda
    L.check(L.lib().acx_bn_running_update(h, mean.data_ptr(), var_u.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                          nbt.data_ptr(), mean.numel(), float(bn.momentum), float(1.0 - bn.momentum), _stream()), h)


def axpby_(y: torch.Tensor, x: torch.Tensor, a: float, b: float) -> None:
    h = _h(y)
    L.check(L.lib().acx_axpby(h, x.data_ptr(), y.data_ptr(), y.numel(), a, b, _stream()), h)


def colsum(x: torch.Tensor, D: Optional[int] = None, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """deterministic column sums of x[rows, ld] (first D columns); `acc` [D]: acc += sums in the same launch."""
    assert x.dim() == 2 and x.is_contiguous()
    rows, ld = x.shape
    D = D or ld
    if D % 4 == 0 and ld % 4 == 0 and D <= 16384 and x.data_ptr() % 16 == 0 and (acc is None or acc.data_ptr() % 16 == 0):
        # one launch: slab partials + last-arriver reduce in slab order (acx_colsum_fused)
        lib = L.lib()
        nbytes = int(lib.acx_colsum_fused_part_bytes(rows, D))
        part = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=x.device)
        out = acc if acc is not None else torch.empty(D, dtype=torch.float32, device=x.device)
        h = _h(x)
        L.check(lib.acx_colsum_fused(h, x.data_ptr(), ld, rows, D, out.data_ptr(), part.data_ptr(), part.numel() * 4,
                                     _colsum_counters(x.device).data_ptr(), 1.0 if acc is not None else 0.0, _stream()), h)
        return out
    if acc is not None:
        s_ = colsum(x, D)
        axpby_(acc, s_, 1.0, 1.0)
        return acc
    rpb = 128
    nb = (rows + rpb - 1) // rpb
    part = torch.empty(nb, D, dtype=torch.float32, device=x.device)
    h = _h(x)
    L.check(L.lib().acx_colsum_partials(h, x.data_ptr(), ld, part.data_ptr(), rows, D, rpb, _stream()), h)
    return reduce_rows(part)


def text_directions_bwd(text, ncentroid, ddirs, normal_id) -> torch.Tensor:
    Cc, D = text.shape
    dtext = torch.empty_like(text)
    h = _h(text)
    L.check(L.lib().acx_text_directions_bwd(h, text.data_ptr(), ncentroid.data_ptr(), ddirs.data_ptr(), dtext.data_ptr(), Cc, D,
                                            normal_id, _stream()), h)
    return dtext


def selector_dirs_grad(draw, x, ncentroid, text, normal_id, C1: int) -> torch.Tensor:
    """d_text [C, D] from draw [rows, C1 padded] and the features x [rows, D]: d_dirs = draw^T (x - ncentroid) (acx_gemm_tn) and the
    direction normalisation's backward (acx_text_directions_bwd) -- the TN product's K-split partial images are added by the second
    kernel itself (acx_gemm_tn_parts + acx_text_directions_bwd_parts): no reduce launch, the same sums in the same order."""
    assert draw.dim() == 2 and x.dim() == 2 and draw.is_contiguous() and x.is_contiguous() and draw.shape[0] == x.shape[0]
    M, N1 = draw.shape
    D = x.shape[1]
    Cc = text.shape[0]
    assert Cc - 1 == C1 <= N1 and text.shape[1] == D
    lib = L.lib()
    nbytes = lib.acx_gemm_tn_workspace_bytes(M, N1, D)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    out = torch.empty(N1, D, dtype=torch.float32, device=x.device)
    splits = C.c_int32(0)
    h = _h(x)
    L.check(lib.acx_gemm_tn_parts(h, draw.data_ptr(), draw.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(), D, M, N1, D,
                                  _ptr(ncentroid), 0, 0, 0, 0, ws.data_ptr(), ws.numel(), _zero_page(x.device).data_ptr(),
                                  C.byref(splits), _stream()), h)
    dtext = torch.empty_like(text)
    n = int(splits.value)
    src = ws if n > 1 else out
    L.check(lib.acx_text_directions_bwd_parts(h, text.data_ptr(), ncentroid.data_ptr(), src.data_ptr(), n, N1 * D, dtext.data_ptr(), Cc, D,
                                              normal_id, _stream()), h)
    return dtext


def selector_tail(raw, labels, mask_top, mask_bot, N, Lg, normal_id, ktop, kbot, eps, *, gathered=None, stats=None, bn=None):
    """BatchNorm of the selector's raw projections, [running statistics], top-k / bottom-k picks and the gather of the top-k
    segments in ONE launch (acx_selector_tail; bit-identical to selector_bn + bn_running_update_ + select_idx + gather_segments,
    and to bn_combine in front of them when `gathered` [ranks, 2 C1 + 1] is given instead of `stats` = (mean, var_biased,
    var_unbiased)).  Returns (logits, idx_top, idx_bot, logits_topk, (mean, var_biased, var_unbiased, total_rows or None))."""
    rows, C1 = raw.shape
    B = labels.shape[0]
    dev = raw.device
    logits = torch.empty(rows, C1, dtype=torch.float32, device=dev)
    it = torch.empty(B, ktop, dtype=torch.int64, device=dev)
    ib = torch.empty(B, kbot, dtype=torch.int64, device=dev)
    topk = torch.empty(B * ktop * Lg, C1, dtype=torch.float32, device=dev)
    stat_out = None
    if gathered is not None:
        assert gathered.is_cuda and gathered.dtype == torch.float32 and gathered.is_contiguous() and gathered.shape[1] == 2 * C1 + 1
        stat_out = torch.empty(3 * C1 + 1, dtype=torch.float32, device=dev)
        mean = var_b = var_u = None
    else:
        mean, var_b, var_u = stats
    h = _h(raw)
    rm = rv = nbt = None
    mom = 0.0
    if bn is not None:
        rm, rv, nbt, mom = bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)
        assert nbt.dtype == torch.int64 and nbt.is_cuda
    L.check(L.lib().acx_selector_tail(h, raw.data_ptr(), _ptr(gathered), gathered.shape[0] if gathered is not None else 0,
                                      _ptr(mean), _ptr(var_b), _ptr(var_u), _ptr(stat_out), _ptr(rm), _ptr(rv), _ptr(nbt), mom,
                                      float(1.0 - mom), logits.data_ptr(), logits.stride(0), labels.data_ptr(), mask_top.data_ptr(),
                                      mask_bot.data_ptr(), it.data_ptr(), ib.data_ptr(), topk.data_ptr(), B, N, Lg, C1, normal_id, ktop,
                                      kbot, float(eps), _stream()), h)
    if stat_out is not None:
        st = (stat_out[:C1], stat_out[C1:2 * C1], stat_out[2 * C1:3 * C1], stat_out[3 * C1:])
    else:
        st = (mean, var_b, var_u, None)
    return logits, it, ib, topk, st


def mil_loss_bn(sim, sim_topk, labels, scores, idx_topk_abn, idx_topk_nor, idx_bottomk_abn, N, Lg, K, normal_id, lambdas,
                meter: Optional[torch.Tensor] = None, gout: Optional[torch.Tensor] = None):
    """loss + its gradients + `meter += losses` + the scatter of the top-k rows' gradient + the selector BatchNorm's backward sums in
    ONE launch (acx_mil_loss_bn; bit-identical to mil_loss + axpby_ + scatter_segments_ + bn_bwd_stats).  Returns (losses [8],
    dlogits, dscores, bn_sums [2 C1]); raises AcxError(ACX_E_UNSUPPORTED) when B N Lg is not a multiple of 256."""
    import ctypes
    B = labels.shape[0]
    C1 = sim.shape[1]
    dev = sim.device
    dl = torch.empty_like(sim)
    dsc = torch.empty_like(scores)
    losses = torch.empty(8, dtype=torch.float32, device=dev)
    sums = torch.empty(2 * C1, dtype=torch.float32, device=dev)
    nws = (B * N * Lg + 255) // 256 * 8 + (B * K * Lg + 255) // 256
    ws = torch.empty(nws, dtype=torch.float32, device=dev)
    bws = _bn_workspace(sim, sim.shape[0], C1)
    lam = (ctypes.c_float * 7)(*[float(x) for x in lambdas])
    h = _h(sim)
    ctr = _colsum_counters(dev)[_CTR_N - 1:]                # the last arrival counter of the stream's table: the loss's own
    L.check(L.lib().acx_mil_loss_bn(h, sim.data_ptr(), sim_topk.data_ptr(), labels.data_ptr(), scores.data_ptr(),
                                    idx_topk_abn.data_ptr(), idx_topk_nor.data_ptr(), idx_bottomk_abn.data_ptr(), dl.data_ptr(),
                                    dsc.data_ptr(), losses.data_ptr(), _ptr(meter), sums.data_ptr(), ws.data_ptr(), nws, bws.data_ptr(),
                                    bws.numel() * 8, B, N, Lg, C1, K, normal_id, ctypes.cast(lam, ctypes.c_void_p), _ptr(gout),
                                    ctr.data_ptr(), _stream()), h)
    return losses, dl, dsc, sums


def select_idx(logits, labels, mask_top, mask_bot, N, Lg, normal_id, ktop, kbot):
    B = labels.shape[0]
    C1 = logits.shape[-1]
    it = torch.empty(B, ktop, dtype=torch.int64, device=logits.device)
    ib = torch.empty(B, kbot, dtype=torch.int64, device=logits.device)
    h = _h(logits)
    L.check(L.lib().acx_select_idx(h, logits.data_ptr(), labels.data_ptr(), mask_top.data_ptr(), mask_bot.data_ptr(),
                                   it.data_ptr(), ib.data_ptr(), B, N, Lg, C1, normal_id, ktop, kbot, _stream()), h)
    return it, ib


def gather_segments(logits, idx, N, Lg) -> torch.Tensor:
    B, K = idx.shape
    C1 = logits.shape[-1]
    out = torch.empty(B * K * Lg, C1, dtype=torch.float32, device=logits.device)
    h = _h(logits)
    L.check(L.lib().acx_gather_segments(h, logits.data_ptr(), idx.data_ptr(), out.data_ptr(), B, N, Lg, C1, K, _stream()), h)
    return out


def scatter_segments_(dlogits, dout, idx, N, Lg) -> None:
    B, K = idx.shape
    C1 = dlogits.shape[-1]
    h = _h(dlogits)
    L.check(L.lib().acx_scatter_segments(h, dout.data_ptr(), idx.data_ptr(), dlogits.data_ptr(), B, N, Lg, C1, K, _stream()), h)


def mil_loss(sim, sim_topk, labels, scores, idx_topk_abn, idx_topk_nor, idx_bottomk_abn, N, Lg, K, normal_id, lambdas,
             gout: Optional[torch.Tensor] = None):
    """returns (losses[8], dsim, dsim_topk, dscores)."""
    import ctypes
    B = labels.shape[0]
    C1 = sim.shape[1]
    dev = sim.device
    dsim = torch.empty_like(sim)
    dtopk = torch.empty_like(sim_topk)
    dsc = torch.empty_like(scores)
    losses = torch.empty(8, dtype=torch.float32, device=dev)
    nws = (B * N * Lg + 255) // 256 * 8 + (B * K * Lg + 255) // 256
    ws = torch.empty(nws, dtype=torch.float32, device=dev)
    lam = (ctypes.c_float * 7)(*[float(x) for x in lambdas])
    h = _h(sim)
    ctr = _colsum_counters(dev)[_CTR_N - 1:]                # the last arrival counter of the stream's table: the loss's own
    L.check(L.lib().acx_mil_loss_one(h, sim.data_ptr(), sim_topk.data_ptr(), labels.data_ptr(), scores.data_ptr(),
                                     idx_topk_abn.data_ptr(), idx_topk_nor.data_ptr(), idx_bottomk_abn.data_ptr(), dsim.data_ptr(),
                                     dtopk.data_ptr(), dsc.data_ptr(), losses.data_ptr(), ws.data_ptr(), nws, B, N, Lg, C1, K,
                                     normal_id, ctypes.cast(lam, ctypes.c_void_p), _ptr(gout), ctr.data_ptr(), _stream()), h)
    return losses, dsim, dtopk, dsc


def adamw_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step) -> None:
    h = _h(p)
    L.check(L.lib().acx_adamw(h, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                              weight_decay, step, _stream()), h)


def multi_axpy_(ys, xs, a: float = 1.0) -> None:
    """y_i += a * x_i for lists of equally sized contiguous f32 tensors, one launch."""
    n = len(ys)
    if n == 0:
        return
    assert all(y.numel() == x.numel() and y.is_contiguous() and x.is_contiguous() for y, x in zip(ys, xs))
    h = _h(ys[0])
    L.check(L.lib().acx_multi_axpy(h, n, (C.c_void_p * n)(*[y.data_ptr() for y in ys]), (C.c_void_p * n)(*[x.data_ptr() for x in xs]),
                                   (C.c_int64 * n)(*[y.numel() for y in ys]), float(a), _stream()), h)


def adamw_multi_(ps, gs, ms, vs, lrs, wds, beta1, beta2, eps, step) -> None:
    """one launch for all the tensors (lists of equal length; per-tensor lr / weight decay)."""
    n = len(ps)
    if n == 0:
        return
    h = _h(ps[0])
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])           # noqa: E731
    L.check(L.lib().acx_adamw_multi(h, n, arr(ps), arr(gs), arr(ms), arr(vs), (C.c_int64 * n)(*[p.numel() for p in ps]),
                                    (C.c_double * n)(*[float(x) for x in lrs]), (C.c_double * n)(*[float(x) for x in wds]),
                                    beta1, beta2, eps, step, _stream()), h)


def ctx_grad(dx, C, n_ctx, Lc, W, shared, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    shape = (n_ctx, W) if shared else (C, n_ctx, W)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dx.device)
    else:
        assert tuple(out.shape) == shape and out.is_contiguous() and out.dtype == torch.float32
    h = _h(dx)
    L.check(L.lib().acx_ctx_grad(h, dx.data_ptr(), out.data_ptr(), C, n_ctx, Lc, W, int(shared), _stream()), h)
    return out


def scatter_rows(src, idx, rows) -> torch.Tensor:
    out = zeros(rows, src.shape[1], device=src.device)
    h = _h(src)
    L.check(L.lib().acx_scatter_rows(h, src.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), src.shape[1], _stream()), h)
    return out


# ---------------------------------------------------------------------------------- metrics epilogue (section 8f rank 3)
CURVE_RESULT_BYTES = 56


def sort_pairs(keys: torch.Tensor, vals: torch.Tensor, descending: bool = True):
    """Stable radix sort of (f32 key, int32 payload) pairs -> (sorted keys, payload in that order)."""
    assert keys.dtype == torch.float32 and vals.dtype == torch.int32 and keys.is_contiguous() and vals.is_contiguous()
    n = keys.numel()
    assert vals.numel() == n
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    if n == 0:
        return ko, vo
    lib = L.lib()
    ws = torch.empty(int(lib.acx_sort_workspace_bytes(n)), dtype=torch.uint8, device=keys.device)
    h = _h(keys)
    L.check(lib.acx_sort_pairs(h, keys.data_ptr(), vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, int(descending),
                               ws.data_ptr(), ws.numel(), _stream()), h)
    return ko, vo


def sort_pairs_batched(keys: torch.Tensor, vals: torch.Tensor, descending: bool = True):
    """keys [B, n] f32 (rows may be strided), vals [n] (shared by all rows) or [B, n] int32 -> (sorted keys [B, n], payloads [B, n]):
    B stable radix sorts in one launch sequence (acx_sort_pairs_batched)."""
    assert keys.dtype == torch.float32 and vals.dtype == torch.int32 and keys.dim() == 2 and keys.stride(1) == 1
    B, n = keys.shape
    shared = vals.dim() == 1
    assert vals.is_contiguous() and (vals.shape == (n,) if shared else vals.shape == (B, n))
    ko = torch.empty(B, n, dtype=torch.float32, device=keys.device)
    vo = torch.empty(B, n, dtype=torch.int32, device=keys.device)
    if n == 0 or B == 0:
        return ko, vo
    lib = L.lib()
    ws = torch.empty(B * int(lib.acx_sort_workspace_bytes(n)), dtype=torch.uint8, device=keys.device)
    h = _h(keys)
    L.check(lib.acx_sort_pairs_batched(h, keys.data_ptr(), keys.stride(0), vals.data_ptr(), 0 if shared else n, ko.data_ptr(),
                                       vo.data_ptr(), n, B, int(descending), ws.data_ptr(), ws.numel(), _stream()), h)
    return ko, vo


def clf_curve(sorted_scores: torch.Tensor, sorted_labels: torch.Tensor, cls: int, negate: bool,
              result: torch.Tensor, curves: bool = False):
    """Fills one 56-byte acx_curve_result record (`result`: uint8[56] device view); optionally returns the
    (tps, fps, thresholds) arrays (capacity n; the first n_distinct entries are valid)."""
    n = sorted_scores.numel()
    lib = L.lib()
    dev = sorted_scores.device
    ws = torch.empty(int(lib.acx_clf_curve_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    tps = fps = thr = None
    if curves:
        tps = torch.empty(n, dtype=torch.int32, device=dev)
        fps = torch.empty(n, dtype=torch.int32, device=dev)
        thr = torch.empty(n, dtype=torch.float32, device=dev)
    h = _h(sorted_scores)
    L.check(lib.acx_clf_curve(h, sorted_scores.data_ptr(), sorted_labels.data_ptr(), n, cls, int(negate),
                              result.data_ptr(), _ptr(tps), _ptr(fps), _ptr(thr), ws.data_ptr(), ws.numel(),
                              _stream()), h)
    return tps, fps, thr


CURVE_BATCH_MAX = 64      # acx_clf_curve_batched: CurveBatch.cls[64] + a 64-bit negate mask per launch sequence


def clf_curve_batched(sorted_scores: torch.Tensor, sorted_labels: torch.Tensor, cls, negate, results: torch.Tensor,
                      curves: bool = False):
    """B curves in one launch sequence per 64 problems: sorted_scores / sorted_labels [B, n] (dense rows), `cls` / `negate`
    length-B host sequences, `results`: uint8[B * 56] device buffer (record b at byte 56 b).  With curves=True returns the (tps,
    fps, thresholds) arrays of problem 0 (capacity n; the first n_distinct entries are valid).  More than 64 problems (a label
    set with > 63 classes) are cut into groups of 64; only the first group carries the curve arrays."""
    B, n = sorted_scores.shape
    assert sorted_labels.shape == (B, n) and sorted_scores.is_contiguous() and sorted_labels.is_contiguous()
    assert len(cls) == B and len(negate) == B and results.numel() >= B * CURVE_RESULT_BYTES
    lib = L.lib()
    dev = sorted_scores.device
    tps = fps = thr = None
    if curves:
        tps = torch.empty(n, dtype=torch.int32, device=dev)
        fps = torch.empty(n, dtype=torch.int32, device=dev)
        thr = torch.empty(n, dtype=torch.float32, device=dev)
    h = _h(sorted_scores)
    for lo in range(0, B, CURVE_BATCH_MAX):
        hi = min(B, lo + CURVE_BATCH_MAX)
        k = hi - lo
        ws = torch.empty(k * int(lib.acx_clf_curve_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        ca, na = (C.c_int32 * k)(*[int(c) for c in cls[lo:hi]]), (C.c_int32 * k)(*[int(bool(x)) for x in negate[lo:hi]])
        first = lo == 0
        L.check(lib.acx_clf_curve_batched(h, sorted_scores[lo:hi].data_ptr(), n, sorted_labels[lo:hi].data_ptr(), n, n, k, ca, na,
                                          results[lo * CURVE_RESULT_BYTES:].data_ptr(), _ptr(tps) if first else None,
                                          _ptr(fps) if first else None, _ptr(thr) if first else None, ws.data_ptr(), ws.numel(),
                                          _stream()), h)
    return tps, fps, thr


def eval_counts(scores, probs, labels, C: int, normal_idx: int, threshold_dev: torch.Tensor):
    """-> (y_pred int32 [n], counts int64 [3C + C*C + 30]); `threshold_dev` is a device f32 scalar view."""
    n = scores.numel()
    assert probs.shape == (n, C - 1) and probs.is_contiguous() and labels.dtype == torch.int64
    y = torch.empty(n, dtype=torch.int32, device=scores.device)
    counts = torch.empty(3 * C + C * C + 30, dtype=torch.int64, device=scores.device)
    h = _h(scores)
    L.check(L.lib().acx_test_counts(h, scores.data_ptr(), probs.data_ptr(), labels.data_ptr(), n, C, normal_idx,
                                    threshold_dev.data_ptr(), y.data_ptr(), counts.data_ptr(), _stream()), h)
    return y, counts
