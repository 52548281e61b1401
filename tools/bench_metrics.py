"""Times the metrics epilogue (sort + curve + counts) at test-set scale on one GPU, with the oracle beside it.
Usage: python tools/bench_metrics.py [n_frames]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from anomalyclip_amd import metrics as M, ops
from oracle import metrics_oracle as MO

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_111_808        # UCF-Crime test split
C, nid = 14, 7
rng = np.random.default_rng(0)
labels = rng.integers(0, C, n); labels[rng.random(n) < 0.9] = nid
s = np.clip(0.4 * (labels != nid) + 0.6 * rng.random(n), 0, 1).astype(np.float32)
p = rng.random((n, C - 1)).astype(np.float32); p = (p / p.sum(1, keepdims=True) * s[:, None]).astype(np.float32)
ds, dl, dp = (torch.from_numpy(x).cuda() for x in (s, labels, p))
for _ in range(2): M.evaluate(ds, dl, dp, nid, C)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r = M.evaluate(ds, dl, dp, nid, C)
torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / 5
lab32 = dl.to(torch.int32)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10): ops.sort_pairs(ds, lab32)
ev[1].record(); torch.cuda.synchronize()
sort_ms = ev[0].elapsed_time(ev[1]) / 10
# the epilogue's own sort: C + 1 score columns with the shared label payload, one batched launch sequence
allc = torch.rand(C + 1, n, device=ds.device)
ops.sort_pairs_batched(allc, lab32)
ev[0].record()
for _ in range(10): ops.sort_pairs_batched(allc, lab32)
ev[1].record(); torch.cuda.synchronize()
bsort_ms = ev[0].elapsed_time(ev[1]) / 10
t0 = time.perf_counter(); w = MO.epilogue(s, labels, p, nid, C); cpu = time.perf_counter() - t0
print({"n_frames": n, "gpu_epilogue_ms": round(gpu * 1e3, 2), "sort_pairs_ms": round(sort_ms, 3),
       "sort_GBps_alg": round(n * 16 / sort_ms / 1e6, 1), "batched_sort_ms": round(bsort_ms, 3),
       "batched_sort_GBps_alg": round((C + 1) * n * 16 / bsort_ms / 1e6, 1), "cpu_oracle_s": round(cpu, 2),
       "auc_roc": r["auc_roc"], "auc_roc_oracle": w["auc_roc"]})
