#!/bin/bash
# A/B of run-time debug switches on one box: builds ONE -DACX_DEBUG_SWITCHES library and runs the given command under
# each "NAME=VAL[,NAME=VAL...]" setting, interleaved (guide rule 24).   usage: ab_env.sh "<cmd>" set1 set2 ...
cd /root/repo/anomalyclip_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DACX_DEBUG_SWITCHES $ACX_EXTRA_FLAGS acx_api.hip acx_gemm.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip acx_metrics.hip -o /tmp/libacx_dbg.so 2>/dev/null || { echo "build failed"; exit 1; }
cd /root/repo
cmd=$1; shift
for rep in 1 2; do
  for set in "$@"; do
    echo "== $set"
    env ACX_LIB_PATH=/tmp/libacx_dbg.so $(echo $set | tr ',' ' ') $cmd 2>/dev/null
  done
done
