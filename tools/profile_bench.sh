#!/bin/bash
# rocprofv3 captures of bench.py (run on the GPU box through gpurun).  Counter passes are separate runs with
# --kernel-trace only (gpurun refuses --pmc together with sys/hip traces).  Outputs under gpurun_out/prof_bench/.
# usage: profile_bench.sh [auto|f32|bf16]   (auto = the default / headline; -> gpurun_out/prof_bench_<mode>/)
PREC=${1:-auto}
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_bench_$PREC
rm -rf $OUT; mkdir -p $OUT
CMD="python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-live-pmc --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- $CMD > $OUT/$1.log 2>&1; }
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
pmc sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
pmc grbm "GRBM_GUI_ACTIVE"
pmc l2 "TCC_HIT_sum TCC_MISS_sum"
# keep the merge small: drop the per-dispatch traces of the counter passes except the counter tables
find $OUT -name "*_kernel_trace.csv" -path "*/stats/*" -prune -o -name "*_kernel_trace.csv" -print | xargs rm -f
ls -la $OUT $OUT/*
