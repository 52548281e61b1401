cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4x; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xd -o xd -- python $GRAFT_REPO_ROOT/tools/bench_xd.py --head-only --steps 20 > /dev/null 2>&1)
cp $(find /tmp/xd -name '*kernel_stats.csv' | head -1) $O/xd_head_kernel_stats.csv
