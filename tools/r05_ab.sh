#!/bin/bash
# Runs ON THE GPU BOX: the keyword-parallel kernel's time on one config for the product build and variant builds, side by side, twice
# usage: tools/r05_ab.sh <cfg> <variant> ...
export TMPDIR=/tmp
cfg=$1; shift
for rep in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then unset ADCRAFT_HIP_LIB ADCRAFT_ALLOW_STALE_LIB; else export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1; fi
  timeout -k 10 120 python3 tools/exp_sparse_ablation.py $cfg 60 1 2>&1 | tail -1 || exit 1
done
done
