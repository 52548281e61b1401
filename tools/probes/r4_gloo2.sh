cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
ACX_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err
tail -5 $O/bench_gloo2.err
