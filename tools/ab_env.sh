#!/bin/bash
# A/B environment-variable variants of the built library: tools/ab_env.sh "NAME=VAL" ...
for rep in 1 2; do
for spec in "$@"; do
  echo "== $spec"
  env $spec python /root/repo/tools/gemm_bench.py ${GB_ARGS:---shapes qkv,out,fc,proj} 2>/dev/null
done
done
