#!/bin/bash
cd /root/repo/anomalyclip_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DACX_TRACE=1 acx_api.hip acx_gemm.hip acx_norm.hip acx_attn.hip acx_head.hip acx_train.hip -o /tmp/libacx_trace.so 2>/dev/null || { echo build failed; exit 1; }
ACX_LIB_PATH=/tmp/libacx_trace.so python /root/repo/tools/trace_gemm.py 2304 768
ACX_LIB_PATH=/tmp/libacx_trace.so python /root/repo/tools/trace_gemm.py 3072 768
ACX_LIB_PATH=/tmp/libacx_trace.so python /root/repo/tools/trace_gemm.py 768 3072
