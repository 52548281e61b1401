#!/bin/bash
# A/B variants of the GEMM kernel: builds each variant into /tmp and runs the microbench.
# usage: tools/ab_gemm.sh "<name>:<gemm source>:<flags>" ...
cd /root/repo/anomalyclip_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; src=${rest%%:*}; flags=${rest#*:}
  cp $src /tmp/ab_gemm_src.hip; sed -i 's|#include "acx_internal.h"|#include "/root/repo/anomalyclip_amd/csrc/acx_internal.h"|' /tmp/ab_gemm_src.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags acx_api.hip /tmp/ab_gemm_src.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip -o /tmp/libacx_$name.so 2>/dev/null || { echo "build failed $name"; continue; }
done
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name"
  ACX_LIB_PATH=/tmp/libacx_$name.so python /root/repo/tools/gemm_bench.py ${GB_ARGS:---shapes qkv,proj} 2>/dev/null
done
done
