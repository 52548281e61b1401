cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_model.py -x -q -m gpu -k "conv or head or temporal or step_graph_full or gradbuckets_equals or full_config or e2e or config0" 2>&1 | tail -n 12 > $O/pytest.txt
for n in 8 4 1; do python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bh$n.json 2>/dev/null; done
