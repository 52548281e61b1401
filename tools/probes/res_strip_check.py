import sys; sys.path.insert(0, "/root/repo")
import torch
from anomalyclip_amd import ops, _lib as L
torch.manual_seed(0)
for (M, N, K) in ((100864, 768, 768), (100864, 768, 3072), (50433, 772, 544), (33000, 1000, 1024)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    out = ops.gemm(a, w, bias=b, residual=r)
    ref = (a.double() @ w.double().t() + b.double() + r.double())
    err = (out.double() - ref).abs().max().item()
    print(M, N, K, "max err", err, "ref max", ref.abs().max().item())
