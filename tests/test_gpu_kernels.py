"""-m gpu: every libacx kernel against the oracle / a plain fp32 torch CPU reference of the same op.
All calls go through the C ABI (ctypes).  Tolerances are written per test: the PREC_F32 path is an
exact-f32 fma chain (only summation order differs from the CPU), so 1e-5-class tolerances hold;
the bf16 path is checked at bf16 round-off."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from anomalyclip_amd import _lib as L
from anomalyclip_amd import ops
from oracle import anomalyclip_oracle as O

DEV = "cuda"


def relerr(a, b):
    """max |a - b| / max |b| (a NORM-wise relative error, not element-wise: near-zero elements are judged against the
    tensor's scale); the tolerances quoted in the tests are for this quantity."""
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (197 * 3, 2304, 768), (130, 70, 36), (1, 5, 4), (512, 256, 512)])
def test_gemm_f32_plain(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    # asymmetric weights: a transposed C-write cannot pass (guide rule 16)
    w = torch.randn(N, K, generator=g) + torch.arange(N).view(-1, 1) * 0.01
    bias = torch.randn(N, generator=g)
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV))
    ref = a.double() @ w.double().t() + bias.double()
    assert relerr(out, ref) < 2e-6


@pytest.mark.parametrize("M", [1, 31, 77, 154, 320])
@pytest.mark.parametrize("K,N", [(512, 1536), (512, 512), (512, 2048), (2048, 512), (1536, 512), (256, 36), (768, 100)])
def test_gemm_few_rows_kernel(M, K, N):
    """gemm_f32_sk_kernel (acx_gemm with <= 320 rows, K % 256 == 0): the text-tower shapes of a data-parallel rank
    (77 rows per class; qkv / out / fc / proj forward and the transposed dX chain) plus ragged N, every epilogue --
    bias, QuickGELU, residual (also in place), the QuickGELU derivative of a saved pre-activation -- and the QuickGELU
    prologue on A, element-wise against fp64: |err| <= 2e-6 * (sum_k |a||w| + |bias| + |residual|)."""
    g = torch.Generator().manual_seed(M * 7 + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).view(-1, 1) * 1e-4       # asymmetric (guide rule 16)
    bias, res, pre = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ad, wd, bd, rd, pd = (t.to(DEV) for t in (a, w, bias, res, pre))
    a64, w64 = a.double(), w.double()
    lin = a64 @ w64.t() + bias.double()
    bound = 2e-6 * (a64.abs() @ w64.abs().t() + bias.abs().double() + res.abs().double()) + 1e-30

    def ok(out, ref, scale=1.0):
        return bool(((out.cpu().double() - ref).abs() <= bound * scale).all())
    assert ok(ops.gemm(ad, wd, bias=bd), lin)
    assert ok(ops.gemm(ad, wd), lin - bias.double())
    assert ok(ops.gemm(ad, wd, bias=bd, act=L.ACT_QUICKGELU), O.quick_gelu(lin), 1.5)
    assert ok(ops.gemm(ad, wd, bias=bd, residual=rd), lin + res.double())
    x = rd.clone()
    ops.gemm(ad, wd, bias=bd, residual=x, out=x)                                          # C aliases the residual
    assert ok(x, lin + res.double())
    with torch.enable_grad():
        p64 = pre.double().requires_grad_(True)
        O.quick_gelu(p64).backward(torch.ones_like(p64))
    assert ok(ops.gemm(ad, wd, gelu_grad_of=pd), (lin - bias.double()) * p64.grad, 2.0)
    ga = O.quick_gelu(a64)
    bound_g = 2e-6 * (ga.abs() @ w64.abs().t() + bias.abs().double() + res.abs().double()) + 1e-30
    out = ops.gemm(ad, wd, bias=bd, residual=rd, a_act=L.ACT_QUICKGELU)
    assert bool(((out.cpu().double() - (ga @ w64.t() + bias.double() + res.double())).abs() <= 2 * bound_g).all())


@pytest.mark.parametrize("M", [1, 16, 32, 33, 100, 224, 320, 1000])
@pytest.mark.parametrize("N", [1536, 2048, 36])
def test_gemm_few_rows_layernorm_prologue(M, N):
    """a_norm: LayerNorm over K = 512 applied to the A rows inside the few-row kernel (the text tower's ln_1 -> in_proj and
    ln_2 -> c_fc, clip/model.py:214-216) against LayerNorm-then-GEMM in fp64, element-wise; rows with a large common offset
    (|mean| >> std, the case a one-pass variance would lose) included; refused for K != 512 or together with another fusion."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, 512, generator=g)
    a[::3] += 40.0                                           # |mean| = 40 std: two-pass statistics keep the centred values exact
    a[:, 7] *= 25.0                                          # an outlier channel, as CLIP text activations have
    w = torch.randn(N, 512, generator=g) * 0.05 + torch.arange(N).view(-1, 1) * 1e-4
    bias = torch.randn(N, generator=g)
    lw, lb = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g) * 0.1
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), a_norm=(lw.to(DEV), lb.to(DEV)))
    h = torch.nn.functional.layer_norm(a.double(), (512,), lw.double(), lb.double(), 1e-5)
    ref = h @ w.double().t() + bias.double()
    bound = 4e-6 * (h.abs() @ w.double().abs().t() + bias.abs().double()) + 1e-30
    assert bool(((out.cpu().double() - ref).abs() <= bound).all())
    # the unfused pair (LayerNorm launch + GEMM) agrees to round-off
    h32 = ops.layernorm(a.to(DEV), lw.to(DEV), lb.to(DEV))
    assert relerr(out, ops.gemm(h32, w.to(DEV), bias=bias.to(DEV))) < 5e-6
    if M == 32 and N == 36:
        with pytest.raises(L.AcxError, match="a_norm"):
            ops.gemm(a[:, :256].contiguous().to(DEV), w[:, :256].contiguous().to(DEV), a_norm=(lw[:256].to(DEV), lb[:256].to(DEV)))
        with pytest.raises(L.AcxError, match="a_norm"):
            ops.gemm(a.to(DEV), w.to(DEV), a_norm=(lw.to(DEV), lb.to(DEV)), act=L.ACT_QUICKGELU)


def test_gemm_few_row_fusions_are_refused_elsewhere():
    """a_act / gelu_grad_of exist in the few-row kernel only: a shape it cannot take fails loudly, and the row limit is an
    option of the context (above it the tile kernels run and give the same result)."""
    g = torch.Generator().manual_seed(3)
    a, w = torch.randn(64, 96, generator=g).to(DEV), torch.randn(32, 96, generator=g).to(DEV)      # K % 256 != 0
    with pytest.raises(L.AcxError, match="few-row"):
        ops.gemm(a, w, a_act=L.ACT_QUICKGELU)
    a, w = torch.randn(154, 512, generator=g).to(DEV), torch.randn(512, 512, generator=g).to(DEV)
    few = ops.gemm(a, w)
    ops.set_few_row_limit(torch.cuda.current_device(), 0)
    try:
        tiles = ops.gemm(a, w)
    finally:
        ops.set_few_row_limit(torch.cuda.current_device(), 320)
    assert relerr(few, tiles) < 2e-6 and relerr(few, a.double().cpu() @ w.double().cpu().t()) < 2e-6


def test_gemm_epilogues_and_asub():
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 200, 96
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, res, sub = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(K, generator=g)
    base = (a - sub) @ w.t() + bias
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=L.ACT_QUICKGELU, residual=res.to(DEV), a_sub=sub.to(DEV))
    assert relerr(out, O.quick_gelu(base) + res) < 1e-5
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=L.ACT_LEAKYRELU, a_sub=sub.to(DEV))
    assert relerr(out, torch.nn.functional.leaky_relu(base, 0.01)) < 1e-5
    # in-place residual (C aliases residual), as the transformer driver uses it
    x = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), residual=x, out=x)
    assert relerr(x, a @ w.t() + bias + res) < 1e-5


@pytest.mark.parametrize("R,C1", [(1, 13), (2, 13), (8, 17), (3, 64)])
def test_bn_combine_matches_chan_formula(R, C1):
    """acx_bn_combine (SyncBN: the ranks' (mean, biased var x rows, rows) -> statistics over all rows) against the fp64
    evaluation of the same formula (parallel.combine_bn_stats) and against the statistics of the concatenated rows."""
    from anomalyclip_amd import parallel
    g = torch.Generator().manual_seed(R * 100 + C1)
    rows = [int(v) for v in torch.randint(200, 4000, (R,), generator=g)]
    chunks = [torch.randn(n, C1, generator=g, dtype=torch.float64) * (1 + i) + i for i, n in enumerate(rows)]
    gathered = torch.stack([torch.cat([c.mean(0), c.var(0, unbiased=False) * c.shape[0], torch.tensor([float(c.shape[0])], dtype=torch.float64)])
                            for c in chunks])
    m, vb, vu, n = ops.bn_combine(gathered.float().to(DEV).contiguous(), C1)
    allrows = torch.cat(chunks)
    assert float(n) == float(sum(rows))
    assert relerr(m, allrows.mean(0)) < 1e-6 and relerr(vb, allrows.var(0, unbiased=False)) < 1e-5
    assert relerr(vu, allrows.var(0, unbiased=True)) < 1e-5
    rm, rvb, rvu, rn = parallel.combine_bn_stats(gathered[:, :C1], gathered[:, C1:2 * C1], gathered[:, 2 * C1])
    assert relerr(m, rm) < 1e-6 and relerr(vb, rvb) < 1e-5 and relerr(vu, rvu) < 1e-5


@pytest.mark.parametrize("M,N,K,res", [(1078, 512, 2048, True), (1078, 512, 1536, False), (900, 512, 2048, False), (1290, 500, 3072, True)])
def test_gemm_small_tiles_split_k(M, N, K, res):
    """the 64x64-tile kernel with K split 2-4 ways + the fixed-order reduce (long-K narrow outputs above 768 rows: the text
    tower's proj / dX GEMMs of an undivided 14-class batch) against fp64."""
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV) if res else None
    out = ops.gemm(a, w, bias=b, residual=r)
    ref = a.double() @ w.double().t() + b.double() + (r.double() if res else 0.0)
    assert relerr(out, ref) < 2e-6 * (K / 512) ** 0.5
    out2 = ops.gemm(a, w, bias=b, residual=r)
    assert torch.equal(out, out2)                                   # fixed summation order


@pytest.mark.parametrize("cin,cout,tiles", [(64, 256, 2), (256, 64, 1)])
def test_gemm_conv3x3(cin, cout, tiles):
    g = torch.Generator().manual_seed(cin)
    N, Lg = 32, 16
    x = torch.randn(tiles, N, Lg, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = ops.gemm(x.reshape(-1, cin).to(DEV), wk.to(DEV), bias=b.to(DEV), amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=cin)
    assert relerr(out, ref) < 1e-5


@pytest.mark.parametrize("S", [1, 2, 3])
def test_gemm_testtile_and_pos(S):
    g = torch.Generator().manual_seed(S)
    N, Lg, K, E, b = 32, 16, 64, 64, 2
    rows = b * N * S * Lg
    a, w = torch.randn(rows, K, generator=g), torch.randn(E, K, generator=g) * 0.2
    p0, p1 = torch.randn(N, E, generator=g), torch.randn(Lg, E, generator=g)
    src = O.test_tile_index(rows, N, Lg, S)
    ref = (a[src] @ w.t()).view(-1, N, Lg, E) + p0.view(1, N, 1, E) + p1.view(1, 1, Lg, E)
    out = ops.gemm(a.to(DEV), w.to(DEV), amap=L.AMAP_TESTTILE, gn=N, gl=Lg, seg=S, pos0=p0.to(DEV), pos1=p1.to(DEV))
    assert relerr(out, ref.reshape(-1, E)) < 1e-5


def test_gemm_bf16():
    g = torch.Generator().manual_seed(9)
    M, N, K = 394, 768, 768
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    ab, wb = a.bfloat16().float(), w.bfloat16().float()
    ref = ab.double() @ wb.double().t() + bias.double()
    wd = ops.cast_bf16(w.to(DEV))
    assert torch.equal(wd.cpu(), w.bfloat16())           # RNE cast is bit-exact with torch
    out = ops.gemm(a.to(DEV), wd, bias=bias.to(DEV), prec=L.PREC_BF16)            # f32 A converted in the loader
    assert relerr(out, ref) < 1e-5
    out = ops.gemm(ops.cast_bf16(a.to(DEV)), wd, bias=bias.to(DEV), prec=L.PREC_BF16, out_dtype=torch.bfloat16)
    assert relerr(out.float(), ref) < 1e-2


@pytest.mark.parametrize("M,N,K,act,res", [(33001, 2300, 768, 1, 0), (33001, 2300, 768, 0, 1), (70000, 768, 3072, 0, 1),
                                           (100864, 768, 768, 1, 1), (20000, 3072, 768, 0, 0)])
def test_gemm_f32_w8_benchmark_shapes(M, N, K, act, res):
    """The kernel the frame benchmark spends 87 % of its time in -- gemm_f32_w8_kernel<ACT,RES,0> (8 waves, identity rows,
    no split-K: > 256 output tiles, K % 32 == 0) -- at ViT-B/16 sizes with ragged M / N tails, every compile-time
    epilogue, against an fp64 CPU product of the same operands.  (test_gemm_f32_plain's shapes dispatch to the
    64x64-tile / 4-wave kernels.)  Element-wise check: |err| <= 2e-6 * (|a|.|w| + |bias| + |res|) row/col bound."""
    assert ((M + 127) // 128) * ((N + 127) // 128) > 256 and K % 32 == 0
    g = torch.Generator().manual_seed(M + N + K + act + 2 * res)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).view(-1, 1) * 1e-4     # asymmetric (rule 16)
    bias = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=L.ACT_QUICKGELU if act else L.ACT_NONE,
                   residual=r.to(DEV) if res else None).cpu()
    pre = a.double() @ w.double().t() + bias.double()
    ref = O.quick_gelu(pre) if act else pre
    if res:
        ref = ref + r.double()
    # magnitude bound of the f32 accumulation per element: sum_k |a||w| (Cauchy-Schwarz upper bound per row/col pair)
    scale = a.double().norm(dim=1, keepdim=True) * w.double().norm(dim=1).view(1, -1) + bias.double().abs() + 1.0
    err = ((out.double() - ref).abs() / scale).max().item()
    assert err < 2e-6, err
    # max|err| / max|ref|: f32 accumulation round-off grows ~sqrt(K) (guide: 1e-7 .. 3.5e-7 of sum|a.b| for K = 1k .. 4k)
    assert relerr(out, ref) < 2e-6 * max(1.0, K / 768) ** 0.5 * 1.5


@pytest.mark.parametrize("cin,cout,act,res", [(256, 1024, 2, 0), (256, 1024, 0, 0), (128, 512, 2, 0), (1024, 256, 0, 1),
                                              (256, 1024, 0, 1)])
def test_gemm_conv3x3_benchmark_rows(cin, cout, act, res):
    """The implicit-GEMM 3x3 convolutions of the UCF head at benchmark row counts: 33 tiles of 512 tokens = 16 896 rows
    -> 264 ... 1056 output tiles, so NO split-K.  Wide outputs without a residual (>= 1024 tiles of 128 x 128, N >= 512)
    run on the LDS-DMA strip kernel in conv mode -- gemm_f32_p256_kernel<ACT,0,1>: taps outside the grid are DMAs from the
    zero page, two K-steps per tap at cin = 64 ... eight at cin = 256 -- the rest on gemm_f32_w8_kernel<.,.,1> (conv2 +
    residual, N = 256); these are the kernels behind the features/s numbers (the B = 4 training test runs its convs
    through split-K).  Reference: nine shifted fp64 matmuls on the CPU; element-wise bound 3e-6 * sum |x||w|."""
    tiles, N, Lg = (66 if cout == 512 else 33), 32, 16
    M = tiles * N * Lg
    assert ((M + 127) // 128) * ((cout + 127) // 128) > 256
    g = torch.Generator().manual_seed(cin + act)
    x = torch.randn(tiles, N, Lg, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5
    b = torch.randn(cout, generator=g)
    r = torch.randn(M, cout, generator=g) if res else None
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))            # zero halo on the (N, L) grid
    ref = b.double().expand(M, cout).clone()
    mag = b.double().abs().expand(M, cout).clone()
    for kh in range(3):
        for kw in range(3):
            ref += xp[:, kh:kh + N, kw:kw + Lg, :].reshape(M, cin) @ w[:, :, kh, kw].double().t()
            mag += xp[:, kh:kh + N, kw:kw + Lg, :].reshape(M, cin).abs() @ w[:, :, kh, kw].double().abs().t()
    if act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
    if res:
        ref = ref + r.double()
    wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = ops.gemm(x.reshape(-1, cin).to(DEV), wk.to(DEV), bias=b.to(DEV), act=act, residual=r.to(DEV) if res else None,
                   amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=cin)
    assert relerr(out, ref) < 3e-6
    assert bool(((out.cpu().double() - ref).abs() <= 3e-6 * (mag + (r.double().abs() if res else 0))).all())


@pytest.mark.parametrize("M,N,K,act,res,obf", [(20011, 320, 768, 0, 0, 0), (33000, 256, 128, 1, 1, 0),
                                               (16500, 768, 3072, 0, 1, 1), (70000, 128, 64, 1, 0, 1)])
def test_gemm_bf16_lds_dma_path(M, N, K, act, res, obf):
    """bf16 operands resident in global memory + enough tiles to skip split-K -> gemm_bf16_dma_kernel
    (global_load_lds staging, source-side swizzle); ragged M / N tails, every epilogue combination."""
    g = torch.Generator().manual_seed(M + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = a.bfloat16().double() @ w.bfloat16().double().t() + bias.double()
    if act:
        ref = O.quick_gelu(ref)
    if res:
        ref = ref + r.double()
    out = ops.gemm(ops.cast_bf16(a.to(DEV)), ops.cast_bf16(w.to(DEV)), bias=bias.to(DEV), prec=L.PREC_BF16,
                   act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=r.to(DEV) if res else None,
                   out_dtype=torch.bfloat16 if obf else torch.float32)
    assert relerr(out.float(), ref) < (1e-2 if obf else 1e-5)


@pytest.fixture
def ring_everywhere():
    h = L.ctx(0)
    L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, 1), h)
    yield
    L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, 512), h)


@pytest.mark.parametrize("M,N,K,act,res,obf", [(3001, 600, 768, 0, 0, 0), (2500, 512, 128, 0, 1, 0), (5000, 300, 3072, 1, 0, 1),
                                               (777, 1000, 64, 0, 1, 1), (256, 256, 64, 0, 0, 0), (70000, 520, 192, 0, 1, 0)])
def test_gemm_bf16_ring_kernel(M, N, K, act, res, obf, ring_everywhere):
    """persistent 256x256 LDS-DMA kernel (normally only for >= 512 tiles; forced here): ragged tiles, tile counts
    below / above the CU count (several tiles per block), single-step K, residual-as-accumulator-init."""
    g = torch.Generator().manual_seed(M + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = a.bfloat16().double() @ w.bfloat16().double().t() + bias.double()
    if act:
        ref = O.quick_gelu(ref)
    if res:
        ref = ref + r.double()
    out = ops.gemm(ops.cast_bf16(a.to(DEV)), ops.cast_bf16(w.to(DEV)), bias=bias.to(DEV), prec=L.PREC_BF16,
                   act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=r.to(DEV) if res else None,
                   out_dtype=torch.bfloat16 if obf else torch.float32)
    assert relerr(out.float(), ref) < (1e-2 if obf else 1e-5)


@pytest.mark.parametrize("M,N,K", [(1078, 1536, 512), (1078, 512, 2048), (64, 128, 4096), (300, 200, 1024)])
def test_gemm_split_k_paths(M, N, K):
    """skinny problems take the split-K path (workspace handed over by ops.gemm); epilogue applied after the reduce."""
    g = torch.Generator().manual_seed(K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = O.quick_gelu(a.double() @ w.double().t() + bias.double()) + res.double()
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=L.ACT_QUICKGELU, residual=res.to(DEV))
    assert relerr(out, ref) < 3e-6
    x = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), residual=x, out=x)          # in place through the reduce kernel
    assert relerr(x, a.double() @ w.double().t() + bias.double() + res.double()) < 3e-6


def test_gemm_errors():
    a, w = torch.randn(8, 6, device=DEV), torch.randn(4, 6, device=DEV)
    with pytest.raises(L.AcxError):
        ops.gemm(a, w)                                   # K % 4 != 0
    with pytest.raises(L.AcxError):
        ops.gemm(torch.randn(8, 8), torch.randn(4, 8))   # CPU tensors: no fallback


@pytest.mark.parametrize("D", [64, 128, 256, 512, 768, 1024])
@pytest.mark.parametrize("mode", [L.NORM_LAYER, L.NORM_CHAN])
def test_layernorm(D, mode):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(37, D, generator=g) * 2 + 0.5
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = O.layer_norm(x, w, b) if mode == L.NORM_LAYER else O.chan_layer_norm_last(x, w, b)
    out = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), mode=mode)
    assert relerr(out, ref) < 2e-6
    outb = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), mode=mode, out_dtype=torch.bfloat16)
    assert relerr(outb.float(), ref) < 1e-2


@pytest.mark.parametrize("L_,heads,batch,causal", [(197, 12, 2, False), (77, 8, 3, True), (5, 2, 3, False), (224, 1, 1, False), (33, 2, 1, True),
                                                   # attn16_kernel (non-causal, 129..208): odd / full last chunk, 1- and 2-tile tails
                                                   (129, 2, 3, False), (144, 3, 2, False), (160, 1, 5, False), (177, 2, 2, False),
                                                   (192, 4, 1, False), (193, 1, 2, False), (208, 2, 3, False), (197, 12, 40, False)])
def test_attention(L_, heads, batch, causal):
    g = torch.Generator().manual_seed(L_)
    W = heads * 64
    qkv = torch.randn(batch * L_, 3 * W, generator=g)
    q, k, v = qkv.view(batch, L_, 3, heads, 64).permute(2, 0, 3, 1, 4).double()
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L_, L_), float("-inf"), dtype=torch.float64).triu_(1)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(batch * L_, W)
    out = ops.attention(qkv.to(DEV), batch, L_, heads, causal)
    assert relerr(out, ref) < 3e-6


def test_attention_every_length_both_precisions():
    """acx_attention (f32, causal and not) and acx_attention_bf16 at EVERY sequence length 1 .. 224 (one batch of two sequences
    and three heads each): every kernel choice, every tail / padding case, against fp64."""
    g = torch.Generator().manual_seed(123)
    heads, batch = 3, 2
    W = heads * 64
    worst32, worst16 = 0.0, 0.0
    for L_ in range(1, 225):
        qkv = torch.randn(batch * L_, 3 * W, generator=g)
        q, k, v = qkv.view(batch, L_, 3, heads, 64).permute(2, 0, 3, 1, 4).double()
        s = (q * 0.125) @ k.transpose(-1, -2)
        qd = qkv.to(DEV)
        for causal in ((False, True) if L_ <= 77 or L_ % 16 == 5 else (False,)):
            sm = s + torch.full((L_, L_), float("-inf"), dtype=torch.float64).triu_(1) if causal else s
            ref = (torch.softmax(sm, -1) @ v).transpose(1, 2).reshape(batch * L_, W)
            out = ops.attention(qd, batch, L_, heads, causal)
            e = relerr(out, ref)
            worst32 = max(worst32, e)
            assert e < 3e-6, (L_, causal, e)
        ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(batch * L_, W)
        ob = ops.attention_bf16(qd.to(torch.bfloat16), batch, L_, heads)
        qb = qd.to(torch.bfloat16).double().cpu()
        q2, k2, v2 = qb.view(batch, L_, 3, heads, 64).permute(2, 0, 3, 1, 4)
        refb = (torch.softmax((q2 * 0.125) @ k2.transpose(-1, -2), -1) @ v2).transpose(1, 2).reshape(batch * L_, W)
        eb = relerr(ob.float(), refb)
        worst16 = max(worst16, eb)
        assert eb < 1.5e-2, (L_, eb)
    print("worst relative error: f32", worst32, "bf16", worst16)


@pytest.mark.parametrize("spike_key", [100, 3, 196])
def test_attention_spiked_scores(spike_key):
    """one key dominating one query row (guide rule 26: force the extreme softmax case).  For the online-softmax kernel
    the spike sits in a late chunk (running max jumps, everything accumulated so far is rescaled by ~exp(-100)), in
    the first chunk (later chunks vanish) or in the very last valid key."""
    L_, heads = 197, 1
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(L_, 192, generator=g)
    qkv[7, :64] *= 30.0
    qkv[spike_key, 64:128] = qkv[7, :64] / 30.0 * 3
    q, k, v = qkv.view(1, L_, 3, 1, 64).permute(2, 0, 3, 1, 4).double()
    ref = (torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(L_, 64)
    out = ops.attention(qkv.to(DEV), 1, L_, heads, False)
    assert relerr(out, ref) < 3e-6


@pytest.mark.parametrize("T_axis,e,heads", [(0, 32, 8), (1, 32, 8), (0, 16, 8), (1, 16, 8), (0, 32, 2), (1, 32, 2)])
def test_axial_attention(T_axis, e, heads):
    g = torch.Generator().manual_seed(e + heads)
    tiles, N, Lg = 3, 32, 16
    He = heads * e
    qkv = torch.randn(tiles * N * Lg, 3 * He, generator=g)
    q, k, v = qkv.view(tiles, N, Lg, 3, heads, e).permute(3, 0, 1, 2, 4, 5).double()
    if T_axis == 0:
        q, k, v = (z.transpose(1, 2) for z in (q, k, v))        # (tiles, Lg, N, H, e)
    q, k, v = (z.transpose(2, 3) for z in (q, k, v))            # (tiles, A, H, S, e)
    o = torch.softmax(q @ k.transpose(-1, -2) * e ** -0.5, -1) @ v
    o = o.transpose(2, 3)
    if T_axis == 0:
        o = o.transpose(1, 2)
    ref = o.reshape(tiles * N * Lg, He)
    out = ops.axial_attention(qkv.to(DEV), tiles, N, Lg, heads, e, T_axis)
    assert relerr(out, ref) < 3e-6


def test_selector_kernels():
    g = torch.Generator().manual_seed(3)
    for D, C in ((512, 14), (128, 18), (64, 7)):
        rows, normal_id = 777, 4
        x = torch.randn(rows, D, generator=g) * 0.3 + 0.1
        tf = torch.randn(C, D, generator=g)
        nc = torch.randn(D, generator=g) * 0.1
        rm, rv = torch.randn(C - 1, generator=g) * 0.1, torch.rand(C - 1, generator=g) + 0.1
        dirs = ops.text_directions(tf.to(DEV), nc.to(DEV), normal_id)
        assert relerr(dirs, O.selector_directions(tf, nc, normal_id)) < 2e-6
        raw = ops.selector_project(x.to(DEV), nc.to(DEV), dirs)
        ref_raw = (x - nc).double() @ O.selector_directions(tf, nc, normal_id).double().t()
        assert relerr(raw, ref_raw) < 3e-6
        ev, _, _ = O.selector_logits(x, tf, nc, normal_id, rm, rv, training=False)
        out = ops.selector_bn(raw, rm.to(DEV), rv.to(DEV))
        assert relerr(out, ev) < 1e-5
        m, vb, vu = ops.bn_stats(raw)
        assert relerr(m, ref_raw.mean(0)) < 1e-5 and relerr(vb, ref_raw.var(0, unbiased=False)) < 1e-5
        assert relerr(vu, ref_raw.var(0, unbiased=True)) < 1e-5


@pytest.mark.parametrize("rows,D,C1", [(32768, 512, 13), (16 * 511 + 5, 512, 17), (1000, 512, 6), (40960, 512, 6), (777, 128, 13),
                                      (100, 64, 33), (3, 256, 64), (4099, 256, 40), (2050, 256, 64), (513, 1024, 13), (600, 768, 13), (64, 512, 64), (200, 1024, 33)])
def test_selector_project_mfma_and_fused_stats(rows, D, C1):
    """selector_model.py:54,62,65: the projection as a skinny f32-MFMA GEMM (16-row groups, 1..4 column tiles of 16
    directions, ragged last group, every supported width; (1024, 33) overflows the MFMA kernel's LDS layout and takes the
    wave-per-row kernel) with BatchNorm's batch statistics accumulated in its epilogue, against fp64; the stand-alone two-stage
    statistics and the backward column sums against fp64 as well.  Element-wise: |err| <= 2e-6 * sum_k |x - c||d|."""
    g = torch.Generator().manual_seed(rows + D + C1)
    x = torch.randn(rows, D, generator=g) * 0.3 + 0.1
    nc = torch.randn(D, generator=g) * 0.1
    dirs = torch.nn.functional.normalize(torch.randn(C1, D, generator=g), dim=1)
    ref = (x - nc).double() @ dirs.double().t()
    bound = 2e-6 * ((x - nc).abs().double() @ dirs.abs().double().t()) + 1e-30
    xd, ncd, dd = x.to(DEV), nc.to(DEV), dirs.to(DEV)
    raw = ops.selector_project(xd, ncd, dd)
    assert raw.shape == (rows, C1) and bool(((raw.cpu().double() - ref).abs() <= bound).all())
    raw2, m, vb, vu = ops.selector_project_stats(xd, ncd, dd)
    assert torch.equal(raw2, raw)                                                  # same kernel, same order
    r64 = raw.cpu().double()
    assert relerr(m, r64.mean(0)) < 1e-6 and relerr(vb, r64.var(0, unbiased=False)) < 1e-6
    if rows > 1:
        assert relerr(vu, r64.var(0, unbiased=True)) < 1e-6
    m2, vb2, vu2 = ops.bn_stats(raw)
    assert relerr(m2, r64.mean(0)) < 1e-6 and relerr(vb2, r64.var(0, unbiased=False)) < 1e-6 and relerr(vu2, vu) < 1e-6
    dl = torch.randn(rows, C1, generator=g)
    sums = ops.bn_bwd_stats(raw, dl.to(DEV))
    assert relerr(sums[:C1], dl.double().sum(0)) < 1e-6 and relerr(sums[C1:], (dl.double() * r64).sum(0)) < 1e-6
    # run-to-run identical (fixed-order reductions)
    assert torch.equal(ops.bn_bwd_stats(raw, dl.to(DEV)), sums) and torch.equal(ops.selector_project_stats(xd, ncd, dd)[1], m)


def test_cls_head_class_probs_misc():
    g = torch.Generator().manual_seed(4)
    E, N, Lg, S = 256, 32, 16, 2
    rows = N * Lg * S
    x1, x2 = torch.randn(rows, E, generator=g), torch.randn(rows, E, generator=g)
    lw, lb = torch.randn(E, generator=g), torch.randn(E, generator=g)
    w, b = torch.randn(1, E, generator=g) * 0.1, torch.randn(1, generator=g)
    ref = torch.sigmoid(O.layer_norm((x1 + x2) / 2, lw, lb) @ w.t() + b).view(-1)
    out = ops.cls_head(x1.to(DEV), x2.to(DEV), lw.to(DEV), lb.to(DEV), w.to(DEV), b.to(DEV), N, Lg, 0)
    assert relerr(out, ref) < 1e-5
    src = O.test_tile_index(rows, N, Lg, S)
    out = ops.cls_head(x1.to(DEV), x2.to(DEV), lw.to(DEV), lb.to(DEV), w.to(DEV), b.to(DEV), N, Lg, S)
    exp = torch.empty(rows)
    exp[src] = ref
    assert relerr(out, exp) < 1e-5
    sim, sc = torch.randn(rows, 13, generator=g), torch.rand(rows, generator=g)
    cp, _ = O.eval_postprocess(sim, sc, rows)
    assert relerr(ops.class_probs(sim.to(DEV), sc.to(DEV)), cp) < 1e-5
    acc = torch.zeros(E, device=DEV)
    ops.colsum_(acc, x1.to(DEV))
    assert relerr(acc, x1.double().sum(0)) < 1e-5


@pytest.mark.parametrize("batch,L,heads", [(3, 197, 12), (2, 77, 8), (5, 32, 2), (1, 224, 1), (4, 50, 3)])
def test_attention_bf16(batch, L, heads):
    """bf16 attention kernel (bf16 mode) vs fp64 softmax attention on the bf16-rounded inputs."""
    g = torch.Generator().manual_seed(L)
    W = heads * 64
    qkv = (torch.randn(batch * L, 3 * W, generator=g) * 0.8).bfloat16()
    x = qkv.double().view(batch, L, 3, heads, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(batch * L, W)
    out = ops.attention_bf16(qkv.to(DEV), batch, L, heads)
    assert out.dtype == torch.bfloat16
    assert relerr(out.float(), ref) < 1.5e-2


@pytest.mark.parametrize("M,cin,cout,act,res", [(512, 256, 1024, L.ACT_LEAKYRELU, False), (512, 1024, 256, L.ACT_NONE, True),
                                                (1024, 64, 128, L.ACT_NONE, False)])
def test_conv_gemm_split_k(M, cin, cout, act, res):
    """single-video conv GEMMs (8-16 output tiles, K up to 9216) take the split-K path of the 8-wave kernel;
    LeakyReLU / residual are applied by the reduce kernel."""
    g = torch.Generator().manual_seed(M + cin)
    N, Lg = 32, 16
    x = torch.randn(M, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5
    b, r = torch.randn(cout, generator=g), torch.randn(M, cout, generator=g)
    xi = x.view(-1, N, Lg, cin).permute(0, 3, 1, 2).double()
    ref = torch.nn.functional.conv2d(xi, w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(M, cout)
    if act == L.ACT_LEAKYRELU:
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
    if res:
        ref = ref + r.double()
    wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = ops.gemm(x.to(DEV), wk.to(DEV), bias=b.to(DEV), act=act, residual=r.to(DEV) if res else None,
                   amap=L.AMAP_CONV3X3, gn=N, gl=Lg, cin=cin)
    assert relerr(out, ref) < 1e-5


def test_round3_gemm_kernels_random_shapes():
    """Seeded random shapes through the three GEMM kernels added in round 3, element-wise against fp64 (bound
    2e-6 ... 3e-6 * sum |a||w|): the few-row kernel (M <= 1280 with N <= 512, or M <= 320; K a multiple of 256; ragged N),
    the 256 x 256 weight-gradient kernel (ragged M / N1 / N2) and the strip kernel's conv mode (ragged tile counts)."""
    rng = np.random.default_rng(2026)
    g = torch.Generator().manual_seed(2026)
    for _ in range(10):                                             # few-row
        wide = bool(rng.integers(0, 2))
        M = int(rng.integers(1, 321)) if wide else int(rng.integers(1, 1281))
        N = 4 * int(rng.integers(1, 600)) if wide else 4 * int(rng.integers(1, 129))
        K = 256 * int(rng.integers(1, 9))
        a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
        res = torch.randn(M, N, generator=g)
        out = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), residual=res.to(DEV))
        ref = a.double() @ w.double().t() + b.double() + res.double()
        bound = 2e-6 * (a.double().abs() @ w.double().abs().t() + b.abs().double() + res.abs().double())
        assert bool(((out.cpu().double() - ref).abs() <= bound).all()), ("few-row", M, N, K)
    for _ in range(4):                                              # weight gradient, 256 x 256 tiles
        M = int(rng.integers(4096, 20000))
        N1, N2 = 4 * int(rng.integers(48, 280)), 4 * int(rng.integers(256, 640))
        if ((N1 + 255) // 256) * ((N2 + 255) // 256) < 8:
            N2 = 4 * 640
        a, bm = torch.randn(M, N1, generator=g), torch.randn(M, N2, generator=g)
        out = ops.gemm_tn(a.to(DEV), bm.to(DEV))
        ref = a.double().t() @ bm.double()
        assert bool(((out.cpu().double() - ref).abs() <= 2e-6 * (a.double().abs().t() @ bm.double().abs())).all()), ("tn", M, N1, N2)
    for _ in range(3):                                              # strip kernel, conv mode
        cin, cout = int(rng.choice([64, 128, 256])), int(rng.choice([512, 768, 1024]))
        tiles = int(np.ceil(1024 / (4 * (cout // 128)))) + int(rng.integers(0, 9))      # >= 1024 tiles of 128 x 128
        N_, Lg = 32, 16
        x = torch.randn(tiles, N_, Lg, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5
        bias = torch.randn(cout, generator=g)
        xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
        M = tiles * N_ * Lg
        ref = bias.double().expand(M, cout).clone()
        mag = bias.double().abs().expand(M, cout).clone()
        for kh in range(3):
            for kw in range(3):
                xs = xp[:, kh:kh + N_, kw:kw + Lg, :].reshape(M, cin)
                ref += xs @ w[:, :, kh, kw].double().t()
                mag += xs.abs() @ w[:, :, kh, kw].double().abs().t()
        wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
        out = ops.gemm(x.reshape(-1, cin).to(DEV), wk.to(DEV), bias=bias.to(DEV), act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3,
                       gn=N_, gl=Lg, cin=cin)
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
        assert bool(((out.cpu().double() - ref).abs() <= 3e-6 * mag).all()), ("conv", cin, cout, tiles)


@pytest.mark.parametrize("M,N,K,act,res", [(2048, 768, 768, 0, 1), (1536, 2304, 768, 0, 0), (1024, 3072, 768, 1, 0), (1000, 772, 3072, 0, 1),
                                           (257, 256, 128, 0, 0)])
def test_gemm_bf16x6_is_f32_accurate(M, N, K, act, res):
    """acx_gemm_desc.pairs = 6: A and W as three bf16 planes each (acx_split_bf16x3: exact 24-bit split), the six leading cross
    products on the bf16 matrix cores with f32 accumulation.  Against fp64: element-wise within 2e-6 * sum_k |a||w| (the bound
    the f32 MFMA kernels are held to), and no worse than 1.5x the f32 MFMA kernel's own maximum error on the same operands --
    including a massive-activation column and operands spanning 12 binades."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())
    a[:, 3] *= 60.0
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    x = torch.randn(M, N, generator=g) if res else None
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    xd = x.to(DEV) if res else None
    a3, w3 = ops.split_bf16x3(ad), ops.split_bf16x3(wd)
    assert torch.equal(a3.float().sum(0), ad) and torch.equal(w3.float().sum(0), wd)      # hi + mid + lo == x exactly
    h = L.ctx(torch.cuda.current_device())
    L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, 1), h)
    try:
        y6 = ops.gemm_x6(a3, w3, bias=bd, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=xd)
    finally:
        L.check(L.lib().acx_set_option(h, L.OPT_RING_MIN_TILES, 512), h)
    y32 = ops.gemm(ad, wd, bias=bd, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=xd)
    pre = a.double() @ w.double().t() + b.double()
    ref = pre * torch.sigmoid(1.702 * pre) if act else pre
    if res:
        ref = ref + x.double()
    bound = 2e-6 * (a.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    e6, e32 = (y6.cpu().double() - ref).abs(), (y32.cpu().double() - ref).abs()
    assert bool((e6 <= bound).all()), float((e6 / bound).max())
    assert float(e6.max()) <= 1.5 * float(e32.max()) + 1e-12
    # a single row tile (the plane-reuse kernel takes any size; K split across workgroups through the module's workspace)
    ys = ops.gemm_x6(a3[:, :64].contiguous(), w3, bias=bd)
    es = (ys.cpu().double() - (a[:64].double() @ w.double().t() + b.double())).abs()
    assert bool((es <= 2e-6 * (a[:64].double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30).all())
    with pytest.raises(L.AcxError):                                      # K not a multiple of the 32-wide K-step: refused
        ops.gemm_x6(a3[:, :, :40].contiguous(), w3[:, :, :40].contiguous(), bias=bd)


@pytest.mark.parametrize("M,N,K,act,res,panels", [(2048, 768, 768, 0, 1, 0), (1536, 2304, 768, 0, 0, 3), (1024, 3072, 768, 1, 0, 3), (1000, 772, 3072, 0, 1, 0),
                                                  (70144, 768, 768, 0, 1, 3), (64, 768, 3072, 0, 0, 3)])
def test_gemm_three_products_error_and_bits(M, N, K, act, res, panels):
    """acx_gemm_desc.pairs = 3 (opt-in precision "bf16x3": the three leading cross products of the plane split only; NOT an f32-accurate
    path): against fp64 element-wise within 3e-5 * sum_k |a||w| (dropped terms <= 2^-16 of the leading one, three of them), at least an
    order of magnitude below a plain bf16 product, for row-major and K-panel planes, whole tiles, column strips (70144 rows = 822 tiles:
    a partly filled last round) and a K-split single row tile; run-to-run identical; the lo planes are never read (filled with NaN)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())
    a[:, 3] *= 60.0
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    x = torch.randn(M, N, generator=g) if res else None
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    xd = x.to(DEV) if res else None
    a3, w3 = ops.split_bf16x3(ad, panel=bool(panels)), ops.split_bf16x3(wd, panel=bool(panels))
    a3[2].fill_(float("nan"))
    w3[2].fill_(float("nan"))
    kw = dict(bias=bd, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=xd, panels=panels, pairs=3)
    y3 = ops.gemm_x6(a3, w3, **kw)
    assert torch.isfinite(y3).all()
    assert torch.equal(y3, ops.gemm_x6(a3, w3, **kw))
    if M > 8192:                                        # (fp64 reference on the GPU for the big case)
        pre = ad.double() @ wd.double().t() + bd.double()
        mag = ad.double().abs() @ wd.double().abs().t() + bd.double().abs()
        ref = pre + (xd.double() if res else 0.0)
        e3 = (y3.double() - ref).abs()
    else:
        pre = a.double() @ w.double().t() + b.double()
        mag = a.double().abs() @ w.double().abs().t() + b.double().abs()
        ref = pre * torch.sigmoid(1.702 * pre) if act else pre
        if res:
            ref = ref + x.double()
        e3 = (y3.cpu().double() - ref).abs()
    assert bool((e3 <= 3e-5 * mag + 1e-30).all()), float((e3 / mag).max())
    print("pairs = 3: max err / sum|a||w|", float((e3 / mag).max()))
    if not act and M <= 8192 and N % 8 == 0:
        # plane output of the three-product mode: hi and mid planes only; their sum is the f32 result to 2^-16
        o3 = ops.gemm_x6(a3, w3, bias=bd, panels=panels, pairs=3, planes_out=True, panel_out=bool(panels))
        two = (ops.unpanel(o3) if panels else o3)[:2].float().sum(0)
        yb = ops.gemm_x6(a3, w3, bias=bd, panels=panels, pairs=3)
        assert bool(((two - yb).abs() <= 2.0 ** -15 * yb.abs() + 1e-30).all())


@pytest.mark.parametrize("M,N,K,act,res", [(2048, 768, 768, 0, 1), (1536, 2304, 768, 0, 0), (1024, 3072, 768, 1, 0), (1000, 772, 3072, 0, 1),
                                           (70144, 768, 768, 0, 1), (64, 768, 3072, 0, 0)])
def test_gemm_f16x3_is_f32_accurate(M, N, K, act, res):
    """The ACX_PREC_F16X3 product (opt-in precision "f16x3"): A and W as TWO fp16 planes each (acx_split_f16x2: hi = fp16(x),
    lo = fp16(x - hi); the weights' planes hold 2^10 w, undone by out_scale), the three products (lo,hi) (hi,lo) (hi,hi) on
    v_mfma_f32_32x32x16_f16 with f32 accumulation.  Inside fp16's range -- rows spanning 8 binades around 1, a x60 massive-activation
    column -- against fp64: element-wise within 2e-6 * sum_k |a||w| (THE bound of the f32 MFMA kernels and of the six-product default)
    and no worse than 1.5 x the f32 MFMA kernel's own maximum error; the split reproduces the operands to 2^-23; whole tiles, column
    strips (70144 rows), a K-split single row tile; run-to-run identical."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-3, 5, (M, 1), generator=g).float())
    a[:, 3] *= 60.0
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    x = torch.randn(M, N, generator=g) if res else None
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    xd = x.to(DEV) if res else None
    a2, w2 = ops.split_f16x2(ad, panel=True), ops.split_f16x2(wd, panel=True, scale=1024.0)
    ra, rw = ops.unpanel(a2).float().sum(0), ops.unpanel(w2).float().sum(0) / 1024.0
    assert bool(((ra - ad).abs() <= 2.0 ** -23 * ad.abs() + 2.0 ** -24).all()) and bool(((rw - wd).abs() <= 2.0 ** -23 * wd.abs() + 2.0 ** -24 / 1024.0).all())
    kw = dict(bias=bd, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=xd, panels=3, pairs=3, out_scale=1.0 / 1024.0)
    y = ops.gemm_x6(a2, w2, **kw)
    assert torch.isfinite(y).all() and torch.equal(y, ops.gemm_x6(a2, w2, **kw))
    y32 = ops.gemm(ad, wd, bias=bd, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=xd)
    pre = ad.double() @ wd.double().t() + bd.double()
    mag = ad.double().abs() @ wd.double().abs().t() + bd.double().abs()
    ref = pre * torch.sigmoid(1.702 * pre) if act else pre
    if res:
        ref = ref + xd.double()
    e, e32 = (y.double() - ref).abs(), (y32.double() - ref).abs()
    print("f16x3: max err / sum|a||w|", float((e / mag).max()), " f32 MFMA kernel:", float((e32 / mag).max()))
    assert bool((e <= 2e-6 * mag + 1e-30).all()), float((e / mag).max())
    assert float(e.max()) <= 1.5 * float(e32.max()) + 1e-12
    if not act and N % 8 == 0 and M >= 256:
        # fp16 plane output (K-panel layout): hi + lo reproduces the f32 result of the same product to 2^-22
        o2 = ops.gemm_x6(a2, w2, bias=bd, panels=3, pairs=3, out_scale=1.0 / 1024.0, planes_out=True, panel_out=True)
        yb = ops.gemm_x6(a2, w2, bias=bd, panels=3, pairs=3, out_scale=1.0 / 1024.0)
        two = ops.unpanel(o2).float().sum(0)
        assert bool(((two - yb).abs() <= 2.0 ** -22 * yb.abs() + 2.0 ** -24).all())


def test_gemm_f16x3_outside_its_operand_range():
    """What precision "f16x3" does OUTSIDE the range it is documented for (the reason it is opt-in):
    (a) rows whose elements are all far below 2^-3 (here ~2^-12): the lo plane is a subnormal fp16 number or zero, every element keeps an
        ABSOLUTE accuracy of 2^-25 only -- the error is bounded by 2e-6 sum|a||w| + 2^-25 sum|w|, i.e. no longer small relative to such
        a row's own result;
    (b) a weight whose 2^10-scaled plane exceeds fp16's largest finite number (|w| >= 63.97): the product is non-finite, loudly --
        VisionTransformer.f16x3_range_report() flags such weights before any launch (tests/test_cpu_host.py)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 768, 768
    a = (torch.randn(M, K, generator=g) * 2.0 ** -12).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.04).to(DEV)
    a2, w2 = ops.split_f16x2(a, panel=True), ops.split_f16x2(w, panel=True, scale=1024.0)
    y = ops.gemm_x6(a2, w2, panels=3, pairs=3, out_scale=1.0 / 1024.0)
    ref = a.double() @ w.double().t()
    mag = a.double().abs() @ w.double().abs().t()
    floor = 2.0 ** -25 * w.double().abs().sum(1)[None, :]
    e = (y.double() - ref).abs()
    assert torch.isfinite(y).all() and bool((e <= 2e-6 * mag + floor).all())
    print("tiny rows: max err / max|ref|", float(e.max() / ref.abs().max()), "(six bf16 products on the same operands:",
          float(((ops.gemm_x6(ops.split_bf16x3(a, panel=True), ops.split_bf16x3(w, panel=True), panels=3).double() - ref).abs().max() / ref.abs().max())), ")")
    wbig = w.clone()
    wbig[5, 7] = 100.0
    yb = ops.gemm_x6(ops.split_f16x2((a * 2.0 ** 12).contiguous(), panel=True), ops.split_f16x2(wbig, panel=True, scale=1024.0), panels=3, pairs=3,
                     out_scale=1.0 / 1024.0)
    assert not torch.isfinite(yb[:, 5]).all()            # the overflowing weight's output column is non-finite ...
    assert torch.isfinite(yb[:, :5]).all() and torch.isfinite(yb[:, 6:]).all()     # ... and only that column


@pytest.mark.parametrize("batch,L_,heads", [(3, 197, 12), (2, 208, 2), (40, 197, 12)])
def test_attention_f16x3_vs_fp64(batch, L_, heads):
    """acx_attention_p3n(products = 103): the planes attention on TWO fp16 planes, three exact products per contraction (precision
    "f16x3"): against fp64 no worse than 1.5 x the f32 MFMA attention kernel's maximum error and within 2e-6 of the largest output
    (the six-product default's bounds); identical sequences bit-identical wherever they sit."""
    W = heads * 64
    g = torch.Generator().manual_seed(batch * 1000 + L_)
    qkv = torch.randn(batch * L_, 3 * W, generator=g) * 1.7
    qkv[:, :W] *= 1.5
    if batch > 2:
        qkv[(batch - 1) * L_:] = qkv[:L_]
    qd = qkv.to(DEV)
    q2 = ops.split_f16x2(qd, panel=True)
    o2 = ops.attention_p3(q2, batch, L_, heads, products=103)
    out = ops.unpanel(o2).float().sum(0)
    ref32 = ops.attention(qd, batch, L_, heads, False)
    x = qd.double().view(batch, L_, 3, heads, 64)
    q_, k_, v_ = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q_ @ k_.transpose(-1, -2) / 8.0, dim=-1) @ v_).transpose(1, 2).reshape(batch * L_, W)
    e, e32 = (out.double() - ref).abs().max().item(), (ref32.double() - ref).abs().max().item()
    print("f16x3 attention: max |err| vs fp64", e, " f32 MFMA", e32)
    assert torch.isfinite(out).all() and e <= 1.5 * e32 + 1e-9 and e <= 2e-6 * ref.abs().max().item()
    if batch > 2:
        assert torch.equal(out[(batch - 1) * L_:], out[:L_])


@pytest.mark.parametrize("batch,L_,heads", [(3, 197, 12), (2, 208, 2), (40, 197, 12)])
def test_attention_three_products_vs_fp64(batch, L_, heads):
    """acx_attention_p3n(products = 3): the three leading products of QK^T and PV (precision "bf16x3"); against fp64 within 1e-4 of the
    largest output, the operands' lo planes never read (NaN-filled), identical sequences bit-identical wherever they sit."""
    W = heads * 64
    g = torch.Generator().manual_seed(batch * 1000 + L_)
    qkv = torch.randn(batch * L_, 3 * W, generator=g) * 1.7
    qkv[:, :W] *= 1.5
    if batch > 2:
        qkv[(batch - 1) * L_:] = qkv[:L_]
    qd = qkv.to(DEV)
    q3 = ops.split_bf16x3(qd, panel=True)
    q3[2].fill_(float("nan"))
    o3 = ops.attention_p3(q3, batch, L_, heads, products=3)
    out = ops.unpanel(o3)[:2].float().sum(0)
    x = qd.double().view(batch, L_, 3, heads, 64)
    q_, k_, v_ = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q_ @ k_.transpose(-1, -2) / 8.0, dim=-1) @ v_).transpose(1, 2).reshape(batch * L_, W)
    e3 = (out.double() - ref).abs().max().item()
    print("three products: max |err| / max|ref| vs fp64", e3 / ref.abs().max().item())
    assert torch.isfinite(out).all() and e3 <= 1e-4 * ref.abs().max().item()
    if batch > 2:
        assert torch.equal(out[(batch - 1) * L_:], out[:L_])


@pytest.mark.parametrize("batch,L_,heads", [(3, 197, 12), (2, 208, 2), (1, 193, 1), (40, 197, 12)] + [(3, l_, 3) for l_ in range(194, 208) if l_ != 197])
def test_attention_p3_vs_fp64(batch, L_, heads):
    """acx_attention_p3 (q, k, v as three bf16 planes in K-panel layout; QK^T and PV as bf16 x 6 products, softmax in f32) against
    fp64 softmax attention and against the f32 MFMA kernel on the same inputs: no worse than 1.5 x the f32 kernel's maximum
    error, identical sequences give bit-identical outputs wherever they sit in the batch."""
    W = heads * 64
    g = torch.Generator().manual_seed(batch * 1000 + L_)
    qkv = torch.randn(batch * L_, 3 * W, generator=g) * 1.7
    qkv[:, :W] *= 1.5                                  # logits of a few units: a peaked softmax
    if batch > 2:
        qkv[(batch - 1) * L_:] = qkv[:L_]              # the last sequence repeats the first
    qd = qkv.to(DEV)
    q3 = ops.split_bf16x3(qd, panel=True)
    o3 = ops.attention_p3(q3, batch, L_, heads)
    out = ops.unpanel(o3).float().sum(0)
    ref32 = ops.attention(qd, batch, L_, heads, False)
    x = qd.double().view(batch, L_, 3, heads, 64)
    q_, k_, v_ = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q_ @ k_.transpose(-1, -2) / 8.0, dim=-1) @ v_).transpose(1, 2).reshape(batch * L_, W)
    e3, e32 = (out.double() - ref).abs().max().item(), (ref32.double() - ref).abs().max().item()
    print("max |err| vs fp64: planes", e3, " f32 MFMA", e32)
    assert torch.isfinite(out).all() and e3 <= 1.5 * e32 + 1e-9 and e3 <= 2e-6 * ref.abs().max().item()
    if batch > 2:
        assert torch.equal(out[(batch - 1) * L_:], out[:L_])


def test_split_bf16x3_edge_values():
    """acx_split_bf16x3 at the edges of f32: the split is EXACT (hi + mid + lo == x bit for bit) for every finite x whose lo
    plane stays a normal bf16 number; below that (|x| < 2^-110: the third plane falls under 2^-126) the reconstruction error is
    bounded by bf16's smallest normal, 2^-126, in absolute terms -- f32's own denormal range; values above bf16's largest finite
    number (3.3895e38 .. FLT_MAX) stay finite (hi truncated instead of rounded to infinity); signed zeros keep their sign
    in the hi plane; +-inf and NaN stay non-finite in the hi plane (the products then propagate non-finite values exactly
    where the f32 kernels do: checked through a GEMM row)."""
    tiny = float(2.0 ** -126)
    vals = [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, float(np.float32(3.4028235e38)), -float(np.float32(3.4028235e38)), 3.39e38, 1.17549435e-38, -1.17549435e-38,
            2.0 ** -100, -(2.0 ** -100) * 1.2345678, 2.0 ** -111 * 1.7654321, -(2.0 ** -120) * 1.3333333, 2.0 ** -127, 2.0 ** -140 * 1.5,
            1.401298464e-45, -1.401298464e-45, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 65504.0, 1e-30, -7.7e-20]
    g = torch.Generator().manual_seed(0)
    rnd = (torch.randn(4096, generator=g) * torch.exp2(torch.randint(-126, 121, (4096,), generator=g).float())).tolist()
    x = torch.tensor(vals + rnd + [0.0] * (64 - (len(vals) + len(rnd)) % 64), dtype=torch.float32).view(-1, 64)
    p = ops.split_bf16x3(x.to(DEV)).cpu()
    rec = p[0].double() + p[1].double() + p[2].double()
    xd = x.double()
    big = xd.abs() >= 2.0 ** -100
    assert torch.equal(rec[big].float(), x[big]) and bool((rec[big] == xd[big]).all())          # exact
    assert float((rec - xd).abs().max()) <= tiny                                                 # denormal range: absolute bound
    assert torch.equal(torch.signbit(p[0].float().view(-1)[:2]), torch.tensor([False, True]))    # +0 / -0
    # non-finite values: inf / NaN in A rows 1 and 2, finite row 0 -- outputs are non-finite exactly where the f32 kernel's are
    M, N, K = 256, 256, 256
    a = torch.randn(M, K, generator=g)
    a[1, 7] = float("inf"); a[2, 9] = float("nan"); a[3, 11] = float("-inf")
    w = torch.randn(N, K, generator=g) * 0.1
    ad, wd = a.to(DEV), w.to(DEV)
    a3 = ops.split_bf16x3(ad)
    assert not torch.isfinite(a3[0, 1, 7].float()) and not torch.isfinite(a3[0, 2, 9].float()) and not torch.isfinite(a3[0, 3, 11].float())
    y6 = ops.gemm_x6(a3, ops.split_bf16x3(wd), split_k=False)
    y32 = ops.gemm(ad, wd)
    assert torch.equal(torch.isfinite(y6), torch.isfinite(y32))
    assert bool(torch.isfinite(y6[0]).all()) and bool(torch.isfinite(y6[4:]).all()) and not bool(torch.isfinite(y6[1:4]).any())
    # a product that overflows f32: both paths give +-inf there, equal signs
    a2 = torch.full((M, K), 3.0e19); w2 = torch.full((N, K), 3.0e19); w2[5] = -3.0e19
    z6 = ops.gemm_x6(ops.split_bf16x3(a2.to(DEV)), ops.split_bf16x3(w2.to(DEV)), split_k=False)
    z32 = ops.gemm(a2.to(DEV), w2.to(DEV))
    assert torch.equal(z6, z32) and bool(torch.isinf(z6).all())
    # products of tiny operands against fp64 (absolute bound from the planes' 2^-126 floor)
    a4 = (torch.randn(M, K, generator=g) * 2.0 ** -70)
    w4 = (torch.randn(N, K, generator=g) * 2.0 ** -50)
    y4 = ops.gemm_x6(ops.split_bf16x3(a4.to(DEV)), ops.split_bf16x3(w4.to(DEV)), split_k=False).cpu().double()
    ref4 = a4.double() @ w4.double().t()
    assert float((y4 - ref4).abs().max()) <= 2e-6 * float((a4.double().abs() @ w4.double().abs().t()).max()) + 1e-45


@pytest.mark.parametrize("M,N,K,split", [(512, 256, 2304, True), (1024, 512, 9216, True), (4096, 1024, 2304, True), (300, 260, 96, False), (300, 264, 96, False), (300, 288, 96, False),
                                         (2048, 1024, 768, False), (70000, 768, 768, False)])
def test_gemm_x6_split_k_and_planes_output(M, N, K, split):
    """The plane-reuse kernel (acx_gemm_x6.h): few output tiles -> K split across workgroups + reduce launch, equal to the
    unsplit launch up to summation order; planes output (ACX_BF16X3) of both routes sums back to the f32 output to <= 1 ulp
    of its lo plane; ragged edge tiles; more tiles than CUs (persistent stream across tiles)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    a3, w3 = ops.split_bf16x3(a), ops.split_bf16x3(w)
    y = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_LEAKYRELU, split_k=split)
    y0 = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_LEAKYRELU, split_k=False)
    pre = a.double() @ w.double().t() + b.double()
    ref = torch.where(pre > 0, pre, 0.01 * pre)
    bound = 2e-6 * (a.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    assert bool(((y.double() - ref).abs() <= bound).all()) and bool(((y0.double() - ref).abs() <= bound).all())
    if N % 8 == 0:                                # plane outputs go out as 16-byte pieces of eight bf16 columns
        yp = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_LEAKYRELU, split_k=split, planes_out=True)
        assert yp.shape == (3, M, N) and torch.equal(yp.float().sum(0), y)        # hi + mid + lo == the f32 result exactly
    yb = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_LEAKYRELU, split_k=split, out_dtype=torch.bfloat16)
    assert torch.equal(yb, y.to(torch.bfloat16))
    if K % 32 == 0 and N % 32 == 0:
        # K-panel layout of the planes (ACX_BF16X3P: what the ViT driver's producers write): the same arithmetic in the same order
        a3p, w3p = ops.split_bf16x3(a, panel=True), ops.split_bf16x3(w, panel=True)
        assert torch.equal(ops.unpanel(a3p), a3) and torch.equal(ops.unpanel(w3p), w3)
        assert torch.equal(ops.gemm_x6(a3p, w3p, bias=b, act=L.ACT_LEAKYRELU, split_k=False, panels=3), y0)
        assert torch.equal(ops.gemm_x6(a3, w3p, bias=b, act=L.ACT_LEAKYRELU, split_k=False, panels=2), y0)
        ypp = ops.gemm_x6(a3p, w3p, bias=b, act=L.ACT_LEAKYRELU, split_k=False, panels=3, planes_out=True, panel_out=True)
        assert torch.equal(ops.unpanel(ypp).float().sum(0), y0)
        # ... and the reduce launch of a K split writes the panel layout as well
        yps = ops.gemm_x6(a3p, w3p, bias=b, act=L.ACT_LEAKYRELU, split_k=split, panels=3, planes_out=True, panel_out=True)
        assert torch.equal(ops.unpanel(yps).float().sum(0), y)
    # the multi-tensor split (one launch for a list of dense tensors) writes what the single-tensor launches write
    w3m, a3m = torch.empty_like(w3), torch.empty_like(a3)
    ops.split_bf16x3_multi([(w, w3m), (a, a3m)])
    assert torch.equal(w3m, w3) and torch.equal(a3m, a3)


@pytest.mark.parametrize("rows,D", [(197 * 3, 768), (64, 256), (1001, 512), (7, 1024), (100864, 768)])
def test_layernorm_plane_outputs(rows, D):
    """acx_layernorm with three-plane outputs (the A operand of the ViT's bf16 x 6 in-projection / c_fc): hi + mid + lo is the f32
    LayerNorm bit for bit, and the K-panel layout (two rows per wave, 16-byte stores; odd row counts) holds the row-major planes."""
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 3 + 0.5).to(DEV)
    w, b = (torch.rand(D, generator=g) + 0.5).to(DEV), torch.randn(D, generator=g).to(DEV)
    y = ops.layernorm(x, w, b)
    y3 = ops.layernorm(x, w, b, planes_out=True)
    y3p = ops.layernorm(x, w, b, planes_out=True, panel_out=True)
    assert torch.equal(y3.float().sum(0), y)
    assert torch.equal(ops.unpanel(y3p), y3)


@pytest.mark.parametrize("seed", list(range(12)))
def test_gemm_x6_random_shapes_vs_fp64(seed):
    """Seeded sweep over what acx_gemm takes with pairs = 6: ragged M / N (edge tiles), K = 32 .. 2304 in steps of 32, every
    epilogue (bias, QuickGELU / LeakyReLU, residual; f32, bf16 and plane outputs), K split on or off, the workgroup cap of
    ACX_OPT_X6_CUS, identity rows and the 3x3 convolution on power-of-two grids -- every element against fp64 within the f32
    bound 2e-6 sum_k |a||w|, and the variants of one problem against each other."""
    rng = np.random.RandomState(1000 + seed)
    g = torch.Generator().manual_seed(2000 + seed)
    conv = seed % 3 == 2
    if conv:
        gn, gl = int(rng.choice([4, 8, 32])), int(rng.choice([4, 16]))
        tiles = max(1, 256 // (gn * gl)) * int(rng.randint(1, 4))
        M, cin = tiles * gn * gl, 32 * int(rng.randint(1, 9))
        if M % 256:
            M = (M // 256 + 1) * 256
            tiles = M // (gn * gl)
        K, N = 9 * cin, 4 * int(rng.randint(8, 160))
    else:
        M, N, K = int(rng.randint(1, 1400)), 4 * int(rng.randint(1, 200)), 32 * int(rng.randint(1, 73))
    act = int(rng.choice([L.ACT_NONE, L.ACT_QUICKGELU, L.ACT_LEAKYRELU]))
    res = bool(rng.randint(0, 2)) and act == L.ACT_NONE
    a = (torch.randn(M, K // 9 if conv else K, generator=g) * torch.exp2(torch.randint(-3, 3, (M, 1), generator=g).float())).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV) if res else None
    a3, w3 = ops.split_bf16x3(a), ops.split_bf16x3(w)
    kw = dict(bias=b, act=act, residual=r)
    if conv:
        kw.update(amap=L.AMAP_CONV3X3, gn=gn, gl=gl, cin=K // 9)
        xg = a.double().view(-1, gn, gl, K // 9)
        xp = torch.zeros(xg.shape[0], gn + 2, gl + 2, K // 9, dtype=torch.float64, device=DEV)
        xp[:, 1:-1, 1:-1] = xg
        A = torch.cat([xp[:, kh:kh + gn, kw_:kw_ + gl] for kh in range(3) for kw_ in range(3)], dim=-1).reshape(M, K)
    else:
        A = a.double()
    pre = A @ w.double().t() + b.double()
    ref = pre * torch.sigmoid(1.702 * pre) if act == L.ACT_QUICKGELU else torch.where(pre > 0, pre, 0.01 * pre) if act == L.ACT_LEAKYRELU else pre
    if res:
        ref = ref + r.double()
    bound = 2.5e-6 * (A.abs() @ w.double().abs().t() + b.double().abs() + (r.double().abs() if res else 0)) + 1e-30
    dev = torch.device(DEV).index or 0
    outs = {}
    for split in (True, False):
        for cus in (0, 5):
            try:
                ops.set_x6_cus(dev, cus)
                y = ops.gemm_x6(a3, w3, split_k=split, **kw)
            finally:
                ops.set_x6_cus(dev, 0)
            e = (y.double() - ref).abs()
            assert bool((e <= bound).all()), (seed, M, N, K, conv, act, res, split, cus, float((e / bound).max()))
            outs[(split, cus)] = y
    assert torch.equal(outs[(False, 0)], outs[(False, 5)])                  # without a K split the workgroup count changes nothing
    if not res:
        yb = ops.gemm_x6(a3, w3, out_dtype=torch.bfloat16, **kw)
        assert torch.equal(yb, outs[(True, 0)].to(torch.bfloat16))
        if N % 8 == 0:
            yp = ops.gemm_x6(a3, w3, planes_out=True, **kw)
            assert torch.equal(yp.float().sum(0), outs[(True, 0)])


@pytest.mark.parametrize("M,N,K", [(256 * 70, 1024, 768), (256 * 100 - 37, 768, 1536), (256 * 90, 768, 3072)])
def test_gemm_x6_partial_last_round_is_k_split(M, N, K):
    """More 256 x 256 tiles than CUs with a last round that fills part of the chip (280 / 300 / 270 tiles on 256 CUs): acx_gemm
    sends the whole tile rows of the full rounds out as one launch and the remaining rows as a K-split problem (+ reduce launch
    with the epilogue).  Same product up to the summation order inside the tail rows: the rows of the full rounds are bit-identical
    to the one-launch result, every element is within the f32 bound of fp64; residual in place, bf16 and plane outputs (row-major
    and K-panel: acx_gemm_desc.c_plane_rows) of the split launch hold the f32 result.  Opt-in (ACX_OPT_X6_TAIL_SPLIT)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV)
    a3, w3 = ops.split_bf16x3(a, panel=True), ops.split_bf16x3(w, panel=True)
    y1 = ops.gemm_x6(a3, w3, bias=b, residual=r, split_k=False, panels=3)
    x = r.clone()
    dev = torch.device(DEV).index or 0
    y2_off = ops.gemm_x6(a3, w3, bias=b, residual=r, split_k=True, panels=3)
    assert torch.equal(y2_off, y1)                                                       # opt-in: off by default
    ops.set_x6_tail_split(dev, True)
    try:
        _partial_last_round_checks(M, N, K, a, w, b, r, x, a3, w3, y1)
    finally:
        ops.set_x6_tail_split(dev, False)


def _partial_last_round_checks(M, N, K, a, w, b, r, x, a3, w3, y1):
    y2 = ops.gemm_x6(a3, w3, bias=b, residual=x, out=x, split_k=True, panels=3)          # in place, like the ViT's residual stream
    ncu = torch.cuda.get_device_properties(torch.device(DEV)).multi_processor_count
    tm, tn = (M + 255) // 256, (N + 255) // 256
    row0 = ((tm * tn) // ncu * ncu) // tn * 256
    assert 0 < row0 < M and torch.equal(y2[:row0], y1[:row0]) and not torch.equal(y2[row0:], y1[row0:])   # the tail took the split
    ref = a.double() @ w.double().t() + b.double() + r.double()
    bound = 2e-6 * (a.double().abs() @ w.double().abs().t() + b.double().abs() + r.double().abs()) + 1e-30
    assert bool(((y2.double() - ref).abs() <= bound).all()) and bool(((y1.double() - ref).abs() <= bound).all())
    y = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_QUICKGELU, split_k=True, panels=3)
    yb = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_QUICKGELU, split_k=True, panels=3, out_dtype=torch.bfloat16)
    yp = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_QUICKGELU, split_k=True, panels=3, planes_out=True)
    ypp = ops.gemm_x6(a3, w3, bias=b, act=L.ACT_QUICKGELU, split_k=True, panels=3, planes_out=True, panel_out=True)
    assert torch.equal(yb, y.to(torch.bfloat16)) and torch.equal(yp.float().sum(0), y) and torch.equal(ops.unpanel(ypp), yp)


@pytest.mark.parametrize("M,N,K", [(256 * 70, 1024, 768), (256 * 100 - 37, 768, 1536), (256 * 115, 768, 768), (256 * 197, 768, 768),
                                   (256 * 30 + 5, 768, 3072)])
def test_gemm_x6_strip_tail_bit_identical(M, N, K):
    """A partly filled last round of 256 x 256 tiles goes out as 128- / 64-column STRIPS (the NI = 2 / 1 instantiations of
    gemm_x6_p4_kernel, ACX_OPT_X6_STRIP_TAIL; default: by the cost model).  A strip runs the same K order and product order per
    output element as a whole tile, so -- unlike the K split of the tail -- the result is BIT-IDENTICAL to whole tiles: every
    epilogue the strips carry (f32 plain / QuickGELU / residual in place, plane outputs row-major and K-panel), ragged last row
    tile, forced strip widths, fewer tiles than CUs (30 x 3 = 90 tiles: one round of strips)."""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV)
    a3, w3 = ops.split_bf16x3(a, panel=True), ops.split_bf16x3(w, panel=True)
    dev = torch.device(DEV).index or 0

    def run_all():
        x = r.clone()
        outs = [ops.gemm_x6(a3, w3, bias=b, split_k=False, panels=3),
                ops.gemm_x6(a3, w3, bias=b, act=L.ACT_QUICKGELU, split_k=False, panels=3),
                ops.gemm_x6(a3, w3, bias=b, residual=x, out=x, split_k=False, panels=3).clone(),
                ops.gemm_x6(a3, w3, bias=b, split_k=False, panels=3, planes_out=True),
                ops.gemm_x6(a3, w3, bias=b, act=L.ACT_QUICKGELU, split_k=False, panels=3, planes_out=True, panel_out=True)]
        torch.cuda.synchronize()
        return outs

    ops.set_x6_strip_tail(dev, 0)
    try:
        whole = run_all()
        for mode in (1, 2, 3):
            ops.set_x6_strip_tail(dev, mode)
            for i, (y, y0) in enumerate(zip(run_all(), whole)):
                assert torch.equal(y, y0), (mode, i)
    finally:
        ops.set_x6_strip_tail(dev, 1)
    ref = a.double() @ w.double().t() + b.double()
    bound = 2e-6 * (a.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    assert bool(((whole[0].double() - ref).abs() <= bound).all())
    assert torch.equal(whole[3].float().sum(0), whole[0])


def test_conv3x3_x6_partial_last_round_is_k_split():
    """The same for the implicit 3x3 convolution (a data-parallel rank of two runs its convolutions this way): 280 tiles, the
    tail begins at a token-grid boundary (its taps never leave a grid); rows of the full rounds bit-identical to the one-launch
    result, everything within the f32 bound of fp64."""
    gn, gl, cin, cout, grids = 32, 16, 64, 256, 140
    rows = grids * gn * gl
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(rows, cin, generator=g) * 0.7).to(DEV)
    w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    x3, w3 = ops.split_bf16x3(x), ops.split_bf16x3(w)
    run = lambda: ops.gemm_x6(x3, w3, bias=b, act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3, gn=gn, gl=gl, cin=cin)
    y1 = run()
    dev = torch.device(DEV).index or 0
    ops.set_x6_tail_split(dev, True)
    try:
        y2 = run()
    finally:
        ops.set_x6_tail_split(dev, False)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    row0 = (ncu // 2 * 2) * 256
    assert row0 < rows and torch.equal(y2[:row0], y1[:row0]) and not torch.equal(y2[row0:], y1[row0:])
    xp = torch.zeros(grids, gn + 2, gl + 2, cin, dtype=torch.float64, device=DEV)
    xp[:, 1:-1, 1:-1] = x.double().view(grids, gn, gl, cin)
    cols = torch.cat([xp[:, kh:kh + gn, kw:kw + gl] for kh in range(3) for kw in range(3)], dim=-1).reshape(rows, 9 * cin)
    pre = cols @ w.double().t() + b.double()
    ref = torch.where(pre > 0, pre, 0.01 * pre)
    bound = 2e-6 * (cols.abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    assert bool(((y2.double() - ref).abs() <= bound).all()) and bool(((y1.double() - ref).abs() <= bound).all())


def test_x6_cus_option_caps_the_grid_same_product():
    """ACX_OPT_X6_CUS (ops.set_x6_cus): the persistent pairs = 6 kernels on fewer workgroups, K split chosen for that many -- a
    data-parallel rank leaves CUs to the text stream this way.  Same product up to the summation order of the K pieces (both
    within the f32 bound of fp64), deterministic for a given value, and back to the default's bits when the cap is lifted."""
    gn, gl, cin, cout, tiles = 32, 16, 256, 1024, 8
    rows = tiles * gn * gl
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(rows, cin, generator=g) * 0.7).to(DEV)
    w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    x3, w3 = ops.split_bf16x3(x), ops.split_bf16x3(w)
    run = lambda: ops.gemm_x6(x3, w3, bias=b, act=L.ACT_LEAKYRELU, amap=L.AMAP_CONV3X3, gn=gn, gl=gl, cin=cin)
    y0 = run()
    dev = torch.device(DEV).index or 0
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    try:
        ops.set_x6_cus(dev, ncu - 32)
        y1, y1b = run(), run()
        ops.set_x6_cus(dev, 7)                                             # fewer workgroups than tiles x pieces: several items each
        y2 = run()
    finally:
        ops.set_x6_cus(dev, 0)
    y3 = run()
    assert torch.equal(y1, y1b) and torch.equal(y3, y0)
    xg = x.double().view(tiles, gn, gl, cin)
    xp = torch.zeros(tiles, gn + 2, gl + 2, cin, dtype=torch.float64, device=DEV)
    xp[:, 1:-1, 1:-1] = xg
    cols = torch.cat([xp[:, kh:kh + gn, kw:kw + gl] for kh in range(3) for kw in range(3)], dim=-1).reshape(rows, 9 * cin)
    pre = cols @ w.double().t() + b.double()
    ref = torch.where(pre > 0, pre, 0.01 * pre)
    bound = 2e-6 * (cols.abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    for y in (y0, y1, y2):
        assert bool(((y.double() - ref).abs() <= bound).all())


@pytest.mark.parametrize("cin,cout,tiles,act,res", [(64, 256, 1, 2, 0), (256, 1024, 2, 2, 0), (1024, 256, 8, 0, 1), (128, 512, 40, 0, 0),
                                                     (256, 1024, 64, 2, 0), (512, 128, 8, 0, 1), (512, 128, 64, 0, 0), (512, 128, 1, 0, 1),
                                                     (128, 512, 64, 2, 0), (512, 128, 150, 0, 1)])
def test_conv3x3_x6_vs_f32_conv(cin, cout, tiles, act, res):
    """pairs = 6 with AMAP_CONV3X3 (per-tap source rows, zero page outside the 32 x 16 token grid) against the f32 MFMA
    convolution and fp64: the head's convolutions at 1 .. 64 tiles (K split across workgroups below 256 output tiles); cout = 128
    (the XD-Violence head's c2 and the input gradient of its c1): 256 x 128 tiles, with and without the K split."""
    gn, gl = 32, 16
    rows = tiles * gn * gl
    g = torch.Generator().manual_seed(cin + cout + tiles)
    x = (torch.randn(rows, cin, generator=g) * 0.7).to(DEV)
    w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(rows, cout, generator=g).to(DEV) if res else None
    y32 = ops.gemm(x, w, bias=b, act=act, residual=r, amap=L.AMAP_CONV3X3, gn=gn, gl=gl, cin=cin)
    y6 = ops.gemm_x6(ops.split_bf16x3(x), ops.split_bf16x3(w), bias=b, act=act, residual=r, amap=L.AMAP_CONV3X3, gn=gn, gl=gl, cin=cin)
    # fp64 reference by explicit im2col
    xg = x.double().view(tiles, gn, gl, cin)
    xp = torch.zeros(tiles, gn + 2, gl + 2, cin, dtype=torch.float64, device=DEV)
    xp[:, 1:-1, 1:-1] = xg
    cols = torch.cat([xp[:, kh:kh + gn, kw:kw + gl] for kh in range(3) for kw in range(3)], dim=-1).reshape(rows, 9 * cin)
    pre = cols @ w.double().t() + b.double()
    ref = torch.where(pre > 0, pre, 0.01 * pre) if act == 2 else pre
    if res:
        ref = ref + r.double()
    bound = 2e-6 * (cols.abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    e6, e32 = (y6.double() - ref).abs(), (y32.double() - ref).abs()
    assert bool((e6 <= bound).all()), float((e6 / bound).max())
    assert float(e6.max()) <= 1.5 * float(e32.max()) + 1e-12

