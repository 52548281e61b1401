cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
( time python bench.py > $O/bench_f32.json 2> $O/bench_f32.err ) 2> $O/time.txt
tail -3 $O/bench_f32.err
