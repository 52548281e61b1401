#!/bin/bash
# FETCH_SIZE / L2 hit A/B of GEMM build variants ("<name>:<flags>" ...) on the ViT shapes (run through gpurun)
cd /root/repo/anomalyclip_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags acx_api.hip acx_gemm.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip acx_metrics.hip -o /tmp/libacx_$name.so 2>/dev/null || { echo "build failed $name"; continue; }
done
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}
  for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    out=/tmp/pmc_$name_${ctr%% *}
    rm -rf $out
    ACX_LIB_PATH=/tmp/libacx_$name.so rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python /root/repo/tools/gemm_bench.py --frames 512 --epi 1 --iters 2 > /dev/null 2>&1
    python - "$name" "$out" <<'PY'
import csv, glob, sys, collections
name, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/p_counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "gemm_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
        agg[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(name, k, round(sum(v) / len(v), 1), "n=%d" % len(v))
PY
  done
done
