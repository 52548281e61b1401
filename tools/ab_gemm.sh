#!/bin/bash
# A/B compile-time variants of the GEMM kernel on one box: builds each "<name>:<extra hipcc flags>" into /tmp
# and runs the microbench twice, interleaved (guide rule 24: deltas come from within-probe rounds).
cd /root/repo/anomalyclip_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags acx_api.hip acx_gemm.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip acx_metrics.hip acx_probe.hip -o /tmp/libacx_$name.so 2>/dev/null || { echo "build failed $name"; continue; }
done
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name"
  ACX_LIB_PATH=/tmp/libacx_$name.so python /root/repo/tools/${GB_SCRIPT:-gemm_bench.py} ${GB_ARGS---frames 512} 2>/dev/null
done
done
