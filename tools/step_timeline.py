#!/usr/bin/env python
"""Timeline of ONE training step from a `rocprofv3 --kernel-trace` CSV (development aid): the dispatches between the last two
`adamw_multi_kernel` launches, per queue, with start offset, duration and the idle gap in front of each dispatch --
what the main stream's critical path of a data-parallel rank's step is made of (launch gaps vs kernel time).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/bench_head.py --emulate-world 8 ...
    python tools/step_timeline.py DIR/t_kernel_trace.csv [--delim adamw_multi_kernel] [--out timeline.txt]
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(at::native::)?([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    if name.startswith("at::native") or "at::native" in name[:40]:
        k = re.search(r"(FillFunctor|direct_copy|CatArray|MulFunctor|CUDAFunctor_add|CUDAFunctorOnSelf_add|arange|AUnaryFunctor|BUnaryFunctor)", name)
        return "torch:" + (k.group(1) if k else name[12:60])
    base = name.split("(")[0]
    return base[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--delim", default="adamw_multi")
    ap.add_argument("--out", default=None)
    ap.add_argument("--step", type=int, default=-1, help="which delimited step (default: the last complete one)")
    a = ap.parse_args()
    rows = []
    import gzip
    with (gzip.open(a.trace, "rt") if a.trace.endswith(".gz") else open(a.trace)) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r.get("Stream_Id", "0"),
                         r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if a.delim in r[4]]
    if len(cuts) < 2:
        sys.exit(f"fewer than two '{a.delim}' dispatches in the trace")
    k = a.step if a.step >= 0 else len(cuts) - 2
    lo, hi = cuts[k] + 1, cuts[k + 1] + 1
    seg = rows[lo:hi]
    t0 = rows[cuts[k]][1]                       # end of the previous step's optimizer launch
    out = open(a.out, "w") if a.out else sys.stdout
    last_end = defaultdict(lambda: t0)
    busy = defaultdict(int)
    per_kernel = defaultdict(lambda: [0, 0])
    # union of busy intervals over all queues = time some kernel was running
    ivs = sorted((s, e) for s, e, *_ in seg)
    union, cur_s, cur_e = 0, None, None
    for s, e in ivs:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    print(f"# step {k}: {len(seg)} dispatches, wall {(seg[-1][1] - t0) / 1e3:.1f} us (from the end of the previous optimizer launch "
          f"to the end of this one), some-kernel-running {union / 1e3:.1f} us", file=out)
    print(f"# {'t_us':>9} {'dur_us':>8} {'gap_us':>8}  q/stream  kernel", file=out)
    for s, e, q, st, name in seg:
        key = (q, st)
        gap = s - last_end[key]
        last_end[key] = max(last_end[key], e)
        busy[key] += e - s
        sn = short(name)
        per_kernel[sn][0] += 1
        per_kernel[sn][1] += e - s
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f}  {q}/{st}  {sn}", file=out)
    print("# per queue/stream: busy us", {f"{k[0]}/{k[1]}": round(v / 1e3, 1) for k, v in busy.items()}, file=out)
    print("# per kernel (count, total us), by total:", file=out)
    for sn, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        print(f"#   {c:4d} {t / 1e3:9.1f}  {sn}", file=out)


if __name__ == "__main__":
    main()
