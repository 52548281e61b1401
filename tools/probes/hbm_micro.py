"""rocprofv3 --kernel-trace --stats target: the HBM-bound stragglers in isolation (kernel durations, not host-paced loops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
xs = [torch.randn(32768, 512, generator=g, device=dev) for _ in range(8)]
acc = torch.zeros(512, device=dev)
for i in range(40):
    ops.colsum_(acc, xs[i % 8])
for shape in ((4096, 256), (4096, 1024)):
    ys = [torch.randn(*shape, generator=g, device=dev) for _ in range(8)]
    for i in range(40):
        ops.colsum(ys[i % 8])
ys = [torch.randn(4096, d, generator=g, device=dev) for d in (256, 256, 1024, 256, 1024, 256, 256)]
for i in range(40):
    ops.colsum_group(ys)
nc = torch.zeros(512, device=dev)
dirs = torch.nn.functional.normalize(torch.randn(13, 512, generator=g, device=dev), dim=1)
for i in range(40):
    ops.selector_project_stats(xs[i % 8], nc, dirs)
torch.cuda.synchronize()
