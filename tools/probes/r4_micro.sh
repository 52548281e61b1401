cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mm -o m -- python $GRAFT_REPO_ROOT/tools/probes/hbm_micro.py > /dev/null 2>&1)
cp $(find /tmp/mm -name '*kernel_stats.csv' | head -1) $O/hbm_micro_kernel_stats.csv
