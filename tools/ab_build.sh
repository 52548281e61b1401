#!/bin/bash
# A/B of compile-time variants of libacx on one box: "<cmd>" then "<name>:<extra hipcc flags>" ...; three interleaved rounds.
cmd=$1; shift
cd /root/repo/anomalyclip_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags acx_api.hip acx_gemm.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip acx_metrics.hip -o /tmp/libacx_$name.so 2>/dev/null || echo "build failed $name"
done
cd /root/repo
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}
    echo "== $name"; ACX_LIB_PATH=/tmp/libacx_$name.so $cmd
  done
done
