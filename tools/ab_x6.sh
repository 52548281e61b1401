#!/bin/bash
# A/B of compile-time variants of acx_gemm.hip only (the other objects are the product build's).
#   tools/ab_x6.sh build "<name>:<extra hipcc flags>" ...   (here: hipcc cross-compiles) -> tools/ab_libs/libacx_<name>.so
#   tools/ab_x6.sh run <name> ...                             (GPU box) tools/probes/x6_time.py ${X6_ARGS} twice per variant, interleaved
mode=$1; shift
mkdir -p /root/repo/tools/ab_libs
if [ "$mode" = build ]; then
  cd /root/repo/anomalyclip_amd/csrc
  for spec in "$@"; do
    ( name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c acx_gemm.hip -o /tmp/acx_gemm_$name.o || { echo "build failed $name"; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC acx_api.o /tmp/acx_gemm_$name.o acx_norm.o acx_attn.o acx_head.o acx_train.o acx_metrics.o acx_probe.o acx_step.o -o /root/repo/tools/ab_libs/libacx_$name.so || echo "link failed $name" ) &
  done
  wait
else
  cd /root/repo
  for rep in 1 2; do
  for name in "$@"; do
    echo "== $name"
    ACX_LIB_PATH=/root/repo/tools/ab_libs/libacx_$name.so python tools/probes/x6_time.py ${X6_ARGS:-vit} 2>/dev/null
  done
  done
fi
