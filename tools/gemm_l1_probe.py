"""Probe: is the f32 GEMM limited by the per-CU load path?  Same launch with lda = ldw = 0 (every tile row aliases row 0:
all operand loads hit L1/L2) vs the real strides.  Results are garbage in the aliased run; only time matters."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anomalyclip_amd import _lib as L, ops
M = 197 * 512
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2304, 768)
RES = len(sys.argv) > 3 and sys.argv[3] == "res"
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; out = torch.empty(M, N, device="cuda")
res = torch.randn(M, N, device="cuda") if RES else None
h = L.ctx(0)
def run(lda, ldw, tag):
    d = L.GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = lda, ldw, N
    if RES:
        d.residual, d.ldr = res.data_ptr(), N
    d.a_dtype = d.c_dtype = L.ACX_F32; d.prec = L.PREC_F32
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): L.check(L.lib().acx_gemm(h, C.byref(d), st), h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): L.check(L.lib().acx_gemm(h, C.byref(d), st), h)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{tag:28s} {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
run(K, K, "real strides")
run(0, K, "A rows aliased (lda=0)")
run(K, 0, "W rows aliased (ldw=0)")
run(0, 0, "both aliased")
