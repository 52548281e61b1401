#!/usr/bin/env python
"""Where the HOST spends a replayed training step (development aid): ACX_STEP_HOST_TRACE=1 makes TrainStepGraph.step() stamp
every item it replays; this runs tools/bench_head.py's training loop in-process and prints, for the last few steps, the time
the host spent in each item (a long 'graph' item = hipGraphLaunch blocked or walked a long graph)."""
import os
import sys
import runpy

os.environ["ACX_STEP_HOST_TRACE"] = "1"
sys.argv = ["bench_head.py"] + sys.argv[1:]
here = os.path.dirname(os.path.abspath(__file__))
try:
    runpy.run_path(os.path.join(here, "..", "bench_head.py"), run_name="__main__")
finally:
    from anomalyclip_amd.components import step_graph as sg
    tr = sg._HOST_TRACE or []
    n_timed = int(os.environ.get("TRACE_STEPS", "6"))
    # the timed loop's steps come first after warm-up; print steps 10..10+n of the run
    lo = min(len(tr) - 1, 12)
    t_prev_end = None
    import torch
    torch.cuda.synchronize()
    prev_last = None
    for rec in tr[lo:lo + n_timed]:
        evs = [e for n, e in rec if n == "ev"]
        rec = [r for r in rec if r[0] != "ev"]
        gaps = [f"{evs[i].elapsed_time(evs[i + 1]):.3f}" for i in range(len(evs) - 1)]
        print("device ms between the starts of the main-stream segments (last = optimizer segment incl. what it waited for): " + " ".join(gaps)
              + (f" | previous step's end -> this step's first segment: {prev_last.elapsed_time(evs[0]):.3f}" if prev_last is not None else ""))
        prev_last = evs[-1]
        t0 = rec[0][1]
        if t_prev_end is not None:
            print(f"  (between steps: {1e3 * (t0 - t_prev_end):.3f} ms)")
        line = []
        for (name, t), (_, t_next) in zip(rec[1:-1], rec[2:]):
            line.append(f"{name}={1e3 * (t_next - t):.3f}")
        print(f"step: {1e3 * (rec[-1][1] - t0):.3f} ms host | " + " ".join(line))
        t_prev_end = rec[-1][1]
