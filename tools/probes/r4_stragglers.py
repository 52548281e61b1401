"""HBM-bound stragglers at their benchmark shapes (development probe): selector_project(+stats), colsum over (32768, 512) and the
training-step shapes, preprocess_frames.  Inputs rotate over >= 512 MB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops
from bench import _event_time

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
rows, D, C1 = 64 * 512, 512, 13
xs = [torch.randn(rows, D, generator=g, device=dev) * 0.3 for _ in range(8)]
nc = torch.zeros(D, device=dev)
dirs = torch.nn.functional.normalize(torch.randn(C1, D, generator=g, device=dev), dim=1)
it = [0]


def nxt(lst):
    it[0] += 1
    return lst[it[0] % len(lst)]


def rep(name, nbytes, dt):
    print(f"{name:34s} {dt * 1e6:8.2f} us  {nbytes / dt / 1e12:6.3f} TB/s", flush=True)


for _ in range(2):
    rep("selector_project", rows * D * 4 + rows * C1 * 4, _event_time(lambda: ops.selector_project(nxt(xs), nc, dirs), 48))
    rep("selector_project_stats", rows * D * 4 + rows * C1 * 4, _event_time(lambda: ops.selector_project_stats(nxt(xs), nc, dirs), 48))
    acc = torch.zeros(D, device=dev)
    rep("colsum (32768, 512)", rows * D * 4, _event_time(lambda: ops.colsum_(acc, nxt(xs)), 48))
r0, m0, vb0, vu0 = ops.selector_project_stats(xs[0], nc, dirs)
ref = torch.var_mean(r0.double(), dim=0, unbiased=False)
print("stats max err", float((m0.double() - ref[1]).abs().max()), float((vb0.double() - ref[0]).abs().max()))
print("colsum err", float((ops.colsum(xs[1], D).double() - xs[1].double().sum(0)).abs().max()))
for (r, d) in ((4096, 1024), (4096, 256), (32768, 1024), (32768, 256), (8192, 768)):
    ys = [torch.randn(r, d, generator=g, device=dev) for _ in range(max(2, (1 << 29) // (r * d * 4)))]
    rep(f"colsum ({r}, {d})", r * d * 4, _event_time(lambda: ops.colsum(nxt(ys), d), 48))
    del ys
