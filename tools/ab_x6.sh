#!/bin/bash
# A/B of compile-time variants of ONE source file of libacx (AB_SRC, default acx_gemm.hip; the other objects are the product
# build's).
#   tools/ab_x6.sh build "<name>:<extra hipcc flags>" ...   (here: hipcc cross-compiles) -> tools/ab_libs/libacx_<name>.so
#   tools/ab_x6.sh run <name> ...                             (GPU box) AB_PROBE (default tools/probes/x6_time.py) ${X6_ARGS}, twice per
#                                                             variant, interleaved
mode=$1; shift
SRC=${AB_SRC:-acx_gemm.hip}
mkdir -p /root/repo/tools/ab_libs
if [ "$mode" = build ]; then
  cd /root/repo/anomalyclip_amd/csrc
  others=""
  for o in acx_api acx_gemm acx_norm acx_attn acx_head acx_train acx_metrics acx_probe acx_step acx_comm; do
    [ "$o.hip" = "$SRC" ] || others="$others $o.o"
  done
  for spec in "$@"; do
    ( name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $SRC -o /tmp/ab_$name.o || { echo "build failed $name"; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/ab_$name.o -o /root/repo/tools/ab_libs/libacx_$name.so || echo "link failed $name" ) &
  done
  wait
else
  cd /root/repo
  for rep in 1 2; do
  for name in "$@"; do
    echo "== $name"
    ACX_LIB_PATH=/root/repo/tools/ab_libs/libacx_$name.so python ${AB_PROBE:-tools/probes/x6_time.py} ${X6_ARGS:-vit} 2>/dev/null
  done
  done
fi
