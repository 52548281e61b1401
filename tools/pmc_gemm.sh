#!/bin/bash
# PMC passes for the GEMM microbench (each counter group in its own run; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python /root/repo/tools/gemm_bench.py --iters 3 --shapes $3 > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" qkv,out
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT" qkv,out
run fetch "FETCH_SIZE" qkv,out
run write "WRITE_SIZE" qkv,out
run l2 "TCC_HIT_sum TCC_MISS_sum" qkv,out
ls -R $OUT | head -30
