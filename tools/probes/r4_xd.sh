cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "xd or bf16 or conv" 2>&1 | tail -n 4 > $O/pytest.txt
python tools/bench_xd.py > $O/bench_xd.json 2> $O/bench_xd.err
