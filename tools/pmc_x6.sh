#!/bin/bash
# PMC passes over tools/probes/x6_time.py (each counter group in its own run; kernel-trace only).  $1 = tag, ACX_LIB_PATH selects the build.
cd /tmp && export TMPDIR=/tmp
TAG=${1:-base}
OUT=/root/repo/gpurun_out/pmc_x6_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python /root/repo/tools/probes/x6_time.py ${X6_ARGS:-vit} > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
run sq2 "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
python3 - <<PY
import csv, collections, glob, os
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for d in ("sq1","sq2","grbm"):
    for f in glob.glob(os.path.join(out,d,"**","*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_x6" not in r["Kernel_Name"] and "gemm_bf16_p8" not in r["Kernel_Name"]: continue
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out,d,"**","*kernel_trace.csv"), recursive=True):
        if d!="grbm": continue
        for r in csv.DictReader(open(f)):
            if "gemm_x6" not in r["Kernel_Name"] and "gemm_bf16_p8" not in r["Kernel_Name"]: continue
            dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in agg.items():
    print(k)
    m={c: sum(x)/len(x) for c,x in v.items()}
    for c in sorted(m): print("   %-28s %14.0f  (n=%d)" % (c, m[c], len(v[c])))
    if k in dur:
        du=sum(dur[k])/len(dur[k]); print("   avg duration us %.1f" % du)
        if "GRBM_GUI_ACTIVE" in m: print("   effective clock GHz %.3f" % (m["GRBM_GUI_ACTIVE"]/8/du/1e3))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m: print("   mfma_pipe_busy %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(m["GRBM_GUI_ACTIVE"]/8)))
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_LDS_IDX_ACTIVE","SQ_LDS_BANK_CONFLICT"):
            if c in m: print("   %s / WAVE_CYCLES %.3f" % (c, m[c]/m["SQ_WAVE_CYCLES"]))
PY
