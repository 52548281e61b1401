"""Debug timeline of one GEMM launch (needs a -DACX_TRACE=1 build loaded through ACX_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
M, N, K = 197 * 256, int(sys.argv[1]) if len(sys.argv) > 1 else 2304, int(sys.argv[2]) if len(sys.argv) > 2 else 768
nblk = (M // 128) * (N // 128)
tr = torch.zeros(nblk * 6, dtype=torch.int64, device="cuda")
os.environ["ACX_TRACE_PTR"] = str(tr.data_ptr())
from anomalyclip_amd import ops
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(3):
    ops.gemm(a, w, bias=b, out=out)
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(nblk, 6)
t0 = t[:, 0].min()
start, first, loop_end, end = ((t[:, i] - t0) / 100.0 for i in range(4))     # us (100 MHz clock)
hw = t[:, 4]; xcc = t[:, 5]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
print("blocks", nblk, "kernel span us", end.max())
print("prologue us: mean %.2f  p50 %.2f p90 %.2f" % ((first - start).mean(), np.median(first - start), np.percentile(first - start, 90)))
print("mainloop us: mean %.2f  p50 %.2f p90 %.2f" % ((loop_end - first).mean(), np.median(loop_end - first), np.percentile(loop_end - first, 90)))
print("epilogue us: mean %.2f  p50 %.2f p90 %.2f" % ((end - loop_end).mean(), np.median(end - loop_end), np.percentile(end - loop_end, 90)))
# per-CU slot timeline: group by (xcc, se, sh, cu)
key = xcc * 1000 + se * 100 + sh * 10 + cu
import collections
d = collections.defaultdict(list)
for i in range(nblk):
    d[int(key[i])].append((start[i], end[i], i))
print("distinct CUs", len(d), "blocks per CU min/max", min(len(v) for v in d.values()), max(len(v) for v in d.values()))
k0 = sorted(d)[0]
print("timeline of one CU (start,end,block):")
for s_, e_, i in sorted(d[k0])[:10]:
    print(f"   {s_:8.1f} {e_:8.1f}  dur {e_ - s_:6.1f}  blk {i}")
# gaps: for each CU, sort by start; compute idle = time where fewer than 2 blocks active
busy2 = 0; tot = 0
for v in d.values():
    ev = []
    for s_, e_, _ in v:
        ev += [(s_, 1), (e_, -1)]
    ev.sort()
    act = 0; last = 0
    for tt, dd in ev:
        if act >= 2: busy2 += tt - last
        last = tt; act += dd
    tot += end.max()
print("fraction of CU-time with 2 blocks resident: %.3f" % (busy2 / tot))
