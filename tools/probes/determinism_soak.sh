#!/bin/bash
# Determinism soak of the training step (whole-step graph): 300 optimisation steps, twice, at N = 1 and as rank 0 of an emulated
# 8-rank step -- the final loss must be bit-identical between the two runs (every reduction on the path has a fixed order).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 0 8; do
  for r in 1 2; do
    ARGS="--steps 300 --warmup 2"; [ $n -gt 0 ] && ARGS="$ARGS --emulate-world $n"
    python tools/bench_head.py $ARGS 2>/dev/null | RUN=$r python -c '
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("emulated_world", d.get("emulated_world"), "run", os.environ["RUN"], "loss after 300 steps", repr(d["loss"]), "step_graph", d.get("step_graph"), "ms/step", d["train_ms_per_step"])'
  done
done
