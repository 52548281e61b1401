#!/bin/bash
# UTCL1 (per-CU TLB) counters of the f32 ViT GEMMs: tools/gemm_bench.py under rocprofv3 --pmc, one pass per counter group.
cd /tmp && export TMPDIR=/tmp
for grp in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum" "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"; do
  tag=$(echo $grp | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/tlb_$tag -o p -- python /root/repo/tools/gemm_bench.py --frames 512 --epi 1 --inplace --iters 3 --shapes qkv,out,fc,proj > /tmp/tlb.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/tlb_$tag/**/p_counter_collection.csv", recursive=True)
if not f: print("no output for $grp"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "gemm" not in k: continue
    acc[k[30:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(vs) / len(vs)) for c, vs in d.items()}, "n=", len(next(iter(d.values()))))
PY
done
