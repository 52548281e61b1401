cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "colsum or step_support or misc_train" 2>&1 | tail -3 > $O/pytest.txt
for n in ${WORLDS:-8 1}; do
  python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bh$n.json 2> $O/bh$n.err
  ACX_STEP_SKIP_TEXT=1 python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 > $O/bh${n}_notext.json 2> /dev/null
  python tools/bench_head.py --emulate-world $n --steps 40 --warmup 5 --no-step-graph --text-graph --temporal-graph > $O/bh${n}_autograd.json 2> /dev/null
done
