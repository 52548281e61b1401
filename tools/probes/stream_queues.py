import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anomalyclip_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1])
extra = []
for _ in range(n):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev).add_(1.0)
    extra.append(s)
torch.cuda.synchronize()
main = torch.cuda.current_stream()
import ctypes
from anomalyclip_amd import _lib as L
lib, h = L.lib(), L.ctx(0)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
def times(main, cand):
    tiny = torch.zeros(4, device=dev)
    e0, e1, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        e0.record(main)
        L.check(lib.acx_fill_f32(h, big.data_ptr(), big.numel(), 2.0, main.cuda_stream), h)
        e1.record(main)
    with torch.cuda.stream(cand):
        cand.wait_event(e0)
        L.check(lib.acx_fill_f32(h, tiny.data_ptr(), 4, 1.0, cand.cuda_stream), h)
        ec.record(cand)
    torch.cuda.synchronize()
    return round(e0.elapsed_time(ec), 3), round(e0.elapsed_time(e1), 3)
c0 = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev)
print("extra", n, "main = default stream:", [times(main, c0) for _ in range(3)], " main = a side stream:", [times(side, c0) for _ in range(3)])
for prio in (0, -1):
    cs = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(6)]
    print("extra", n, "prio", prio, "ops.runs_beside vs default:", [int(ops.runs_beside(main, c, dev)) for c in cs], "again:", [int(ops.runs_beside(main, c, dev)) for c in cs])
    cs = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(10)]
    print("extra", n, "prio", prio, "vs default:", [times(main, c)[0] for c in cs])
    print("extra", n, "prio", prio, "vs side   :", [times(side, c)[0] for c in cs])
for prio in ():
    res = []
    cands = []
    for i in range(10):
        c = torch.cuda.Stream(device=dev, priority=prio)
        cands.append(c)
        res.append(int(ops.runs_beside(main, c, dev)))
    print("extra", n, "priority", prio, "runs beside main:", res, " second pass:", [int(ops.runs_beside(main, c, dev)) for c in cands])
