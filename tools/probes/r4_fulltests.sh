cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > $O/pytest_all.txt
