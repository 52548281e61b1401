#!/bin/bash
# PMC passes over tools/attn_bench.py (run on the GPU box through gpurun): separate --pmc runs with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
CMD="python /root/repo/tools/${ATTN_BENCH:-attn_bench.py}"
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- $CMD > $OUT/$1.log 2>&1; }
pmc sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
pmc sq2 "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
pmc grbm "GRBM_GUI_ACTIVE"
python - <<'PY'
import csv, collections, os
out = "/root/repo/gpurun_out/pmc_attn"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq", "sq2", "grbm"):
    p = os.path.join(out, d, "p_counter_collection.csv")
    if not os.path.exists(p):
        print("missing", p); continue
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "attn" not in n: continue
        agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:32s} {sum(xs)/len(xs):16.1f}  (n={len(xs)})")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
        g = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"])
        print("   mfma_pipe_busy_frac", m / 1024.0 / (g / 8.0))
PY
