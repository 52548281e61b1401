#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_ring
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python /root/repo/tools/gemm_bench.py --frames 512 --prec bf16 --iters 3 --shapes qkv,proj --epi 1 > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
run sq2 "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"
run grbm "GRBM_GUI_ACTIVE"
python - <<PY
import csv, collections
for d in ("sq1","sq2","grbm"):
    rows=list(csv.DictReader(open(f"$OUT/{d}/p_counter_collection.csv")))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "ring" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][28:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(d, k, {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()})
PY
