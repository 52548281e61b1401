"""Probe: the text tower's long-K shapes at 1078 rows (M, N = 512, K = 1536 / 2048) through (a) the few-row kernel, (b) the tile
kernels with the split-K workspace ops.gemm hands them, (c) the tile kernels without a workspace (no split)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import _lib as L, ops
h = L.ctx(0)
def run(M, N, K, few, ws_bytes, res=True, n=200):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda"); r = torch.randn(M, N, device="cuda")
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device="cuda")
    ops.set_few_row_limit(0, 1 << 20 if few else 0)
    d = L.GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = M, N, K, K, K, N
    d.a_dtype = d.c_dtype = L.ACX_F32; d.prec = L.PREC_F32; d.bias = b.data_ptr()
    if res: d.residual, d.ldr = r.data_ptr(), N
    if ws_bytes: d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5): L.check(L.lib().acx_gemm(h, C.byref(d), st), h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.check(L.lib().acx_gemm(h, C.byref(d), st), h)
    e1.record(); torch.cuda.synchronize()
    ops.set_few_row_limit(0, 320)
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((1078, 512, 2048), (1078, 512, 1536), (1078, 2048, 512), (1078, 1536, 512), (539, 512, 2048)):
    big = 16 * M * N * 4
    print(f"M={M} N={N} K={K}: few-row {run(M, N, K, True, 0):6.1f} us   tiles+workspace {run(M, N, K, False, big):6.1f} us   "
          f"tiles, no workspace {run(M, N, K, False, 0):6.1f} us")
