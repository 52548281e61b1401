import sys; sys.path.insert(0, "/root/repo")
import torch, bench as B
from anomalyclip_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
rows, D, C1 = 32768, 512, 13
xs = [torch.randn(rows, D, generator=g, device=dev) * 0.3 for _ in range(8)]
nc = torch.zeros(D, device=dev)
dirs = torch.nn.functional.normalize(torch.randn(C1, D, generator=g, device=dev), dim=1)
i = [0]
def f():
    i[0] += 1
    ops.selector_project(xs[i[0] % 8], nc, dirs)
for _ in range(3):
    dt = B._event_time(f, 40)
    print(f"selector_project {dt*1e6:.2f} us  {rows*D*4/dt/1e12:.2f} TB/s")
