cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q -m gpu -k "colsum or grouped or misc_train or ncentroid or step_graph_full or gradbuckets_equals" 2>&1 | tail -n 4 > $O/pytest.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mm -o m -- python $GRAFT_REPO_ROOT/tools/probes/hbm_micro.py > /dev/null 2>&1)
cp $(find /tmp/mm -name '*kernel_stats.csv' | head -1) $O/hbm_micro_kernel_stats.csv
for n in 8 1; do python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bh$n.json 2>/dev/null; done
