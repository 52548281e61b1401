set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4a/pytest.txt
for n in 8 1; do
  D=/tmp/tl$n
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/tools/bench_head.py --emulate-world $n --text-graph --temporal-graph --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r4a/bench_head_w$n.json 2> /tmp/err$n.txt)
  python tools/step_timeline.py $(find $D -name '*kernel_trace.csv' | head -1) --out gpurun_out/r4a/timeline_w$n.txt
done
python tools/bench_head.py --emulate-world 8 --text-graph --temporal-graph --steps 20 --warmup 3 > gpurun_out/r4a/bh8.json 2>&1
python tools/bench_head.py --emulate-world 1 --text-graph --temporal-graph --steps 20 --warmup 3 > gpurun_out/r4a/bh1.json 2>&1
