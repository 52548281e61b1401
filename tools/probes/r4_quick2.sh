cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_km.txt
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "step_graph_full or gradbuckets_equals or full_config or e2e_train" 2>&1 | tail -5 > $O/pytest_tr.txt
for n in 8 1; do python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bh$n.json 2>/dev/null; done
