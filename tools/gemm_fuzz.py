import sys; sys.path.insert(0, "/root/repo")
import torch, random
from anomalyclip_amd import ops, _lib as L
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
CASES = int(sys.argv[2]) if len(sys.argv) > 2 else 14
random.seed(SEED)
torch.manual_seed(SEED)
bad = 0
for it in range(CASES):
    f32 = it % 2 == 0
    M = random.choice([100864, 65536 + random.randrange(1, 4000), 40000 + random.randrange(0, 5000), 131072 + random.randrange(0, 999),
                       16384 + random.randrange(0, 20000), 200000 + random.randrange(0, 3000)])
    N = random.choice([768, 1000, 2304, 2300, 3072, 1536, 516])
    K = random.choice([768, 256, 512, 1024, 3072]) if f32 else random.choice([768, 256, 512, 1024, 3072, 128])
    act = random.choice([0, 1]); res = random.choice([0, 1]) if not act else 0
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05 + torch.arange(N, device="cuda").view(-1, 1) * 1e-4
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if res else None
    if f32:
        out = ops.gemm(a, w, bias=b, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=r)
        ref = a.double() @ w.double().t() + b.double()
        tol = 3e-6 * max(1.0, K / 768) ** 0.5
    else:
        ab, wb = ops.cast_bf16(a), ops.cast_bf16(w)
        out = ops.gemm(ab, wb, bias=b, act=L.ACT_QUICKGELU if act else L.ACT_NONE, residual=r, prec=L.PREC_BF16)
        ref = ab.double() @ wb.double().t() + b.double()
        tol = 2e-5
    if act: ref = ref * torch.sigmoid(1.702 * ref)
    if res: ref = ref + r.double()
    e = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    ok = e < tol
    bad += not ok
    print(("f32 " if f32 else "bf16"), M, N, K, "act", act, "res", res, f"{e:.2e}", "OK" if ok else "FAIL")
    del a, w, ref, out
print("failures:", bad)
