cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -x -q -m gpu -k "few or text or step_graph_full or gradbuckets_equals or world2 or full_config" 2>&1 | tail -n 12 > $O/pytest.txt
for n in 8 1; do python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bh$n.json 2>/dev/null; done
