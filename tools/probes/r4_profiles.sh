cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
for n in 1 2 4 8; do
  python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bench_head_emulated_world$n.json 2> /dev/null
done
python tools/bench_head.py --emulate-world 8 --steps 40 --warmup 5 --no-step-graph > $O/bench_head_emulated_world8_eager.json 2>/dev/null
python tools/bench_head.py --emulate-world 8 --steps 40 --warmup 5 --no-step-graph --text-graph --temporal-graph > $O/bench_head_emulated_world8_autograd_graphs.json 2>/dev/null
ACX_STEP_SKIP_TEXT=1 python tools/bench_head.py --emulate-world 8 --steps 40 --warmup 5 > $O/bench_head_emulated_world8_main_chain_only.json 2>/dev/null
ACX_STEP_SKIP_TEXT=1 python tools/bench_head.py --emulate-world 1 --steps 40 --warmup 5 > $O/bench_head_emulated_world1_main_chain_only.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp8 -o dp8 -- python $GRAFT_REPO_ROOT/tools/bench_head.py --emulate-world 8 --steps 10 --warmup 2 > /dev/null 2>&1)
cp $(find /tmp/pp8 -name '*kernel_stats.csv' | head -1) $O/dp8_rank_share_kernel_stats.csv
python tools/bench_feature_stream.py > $O/feature_stream.json 2>/dev/null
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -2 > $O/pytest_kernels.txt
