#!/bin/bash
# Runs ON THE GPU BOX: binding-budget step times of the product build and of variant builds (adcraft_amd/lib/variants/<name>.so), side by side
# usage: r04_binding_ab.sh <variant> ...
export TMPDIR=/tmp
OUT=gpurun_out/bab; mkdir -p $OUT
for v in product "$@"; do
  if [ $v = product ]; then unset ADCRAFT_HIP_LIB; else export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/exp_binding.py 2>&1 | head -4 || exit 1
done
