cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
python tools/bench_feature_stream.py > $O/feature_stream.json 2> $O/fs.err
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "config0 or feature_stream" 2>&1 | tail -n 3 > $O/pytest_fs.txt
