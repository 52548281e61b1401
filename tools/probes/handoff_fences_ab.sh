#!/bin/bash
# A/B of the last-arriver hand-offs WITH the HIP memory-model fences (-DACX_HANDOFF_FENCES=1, built into tools/ab_libs/libacx_fences.so
# by: hipcc ... -DACX_HANDOFF_FENCES=1 -shared *.hip) against the product library: the training step at N = 1 and as rank 0 of an
# emulated 8-rank step, three interleaved rounds; then the fenced build through the tests that exercise the hand-offs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for lib in product fences; do
    [ $lib = fences ] && export ACX_LIB_PATH=$PWD/tools/ab_libs/libacx_fences.so || unset ACX_LIB_PATH
    for n in 0 8; do
      ARGS="--steps 40 --warmup 5"; [ $n -gt 0 ] && ARGS="$ARGS --emulate-world $n"
      python tools/bench_head.py $ARGS 2>/dev/null | LIB=$lib python -c '
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LIB"], "emulated_world", d.get("emulated_world"), "ms/step", d["train_ms_per_step"], "fwd ms", d["fwd_ms_per_step"], "loss", repr(d["loss"]))'
    done
  done
done
export ACX_LIB_PATH=$PWD/tools/ab_libs/libacx_fences.so
python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -m gpu -k "colsum or mil_loss or loss or few_row or step_graph_full or splitk or split_k" 2>&1 | tail -3
