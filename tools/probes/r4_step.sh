set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "${TESTS:-step or colsum or gradbuckets_equals or text_graph_path or temporal_graph_grad or full_config or layernorm_bwd or misc_train or conv_grads or e2e_train or world2 or gemm_tn}" 2>&1 | tail -40 > $O/pytest_train.txt
for n in ${WORLDS:-8 4 2 1}; do
  timeout 600 python tools/bench_head.py --emulate-world $n --text-graph --temporal-graph --steps 30 --warmup 5 > $O/bh$n.json 2> $O/bh$n.err
done
tail -3 $O/bh8.err
