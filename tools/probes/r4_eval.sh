cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_model.txt
python tools/bench_feature_stream.py > $O/feature_stream.json 2> $O/fs.err
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_model.py 2>&1 | tail -15 > $O/pytest_rest.txt
