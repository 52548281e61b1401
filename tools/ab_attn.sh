# A/B of compile-time variants of the f32 attention kernel: "<name>:<extra hipcc flags>" ...
cd /root/repo/anomalyclip_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DACX_DEBUG_SWITCHES $flags acx_api.hip acx_gemm.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip acx_metrics.hip acx_probe.hip -o /tmp/libacx_$name.so 2>/dev/null || echo "build failed $name"
done
cd /root/repo
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}
    echo -n "$name: "; ACX_LIB_PATH=/tmp/libacx_$name.so python tools/attn_bench.py
  done
done
