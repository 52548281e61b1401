#!/bin/bash
# kernel-time summaries (rocprofv3 --kernel-trace --stats) of the bf16 mode of bench.py, the head benchmark's training
# step and the metrics epilogue; outputs under gpurun_out/prof_extra/ (copy the *_kernel_stats.csv into profiles/).
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_extra
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16 -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --precision bf16 > $OUT/bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python /root/repo/tools/profile_train.py > $OUT/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/metrics -o m -- python /root/repo/tools/bench_metrics.py > $OUT/metrics.log 2>&1
find $OUT -name "*_kernel_trace.csv" | xargs rm -f
ls $OUT/*
# round 2: configs[4] (XD bf16) and rank 0's share of an 8-rank strong-scaling step (emulated on one GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xd -o xd -- python /root/repo/tools/bench_xd.py --steps 4 > $OUT/xd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dp8 -o dp8 -- python /root/repo/tools/bench_head.py --emulate-world 8 --text-graph --temporal-graph --steps 6 > $OUT/dp8.log 2>&1
find $OUT -name "*_kernel_trace.csv" | xargs rm -f
# round 3: rank 0's share of a 2-rank step as well
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dp2 -o dp2 -- python /root/repo/tools/bench_head.py --emulate-world 2 --text-graph --temporal-graph --steps 6 > $OUT/dp2.log 2>&1
find $OUT -name "*_kernel_trace.csv" | xargs rm -f
