#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one kernel on one config for the product build and variant builds
# usage: tools/r05_pmc_ab.sh <kernel substring> <cfg> <variant> ...
export TMPDIR=/tmp
kern=$1; cfg=$2; shift; shift
for v in product "$@"; do
  if [ $v = product ]; then unset ADCRAFT_HIP_LIB ADCRAFT_ALLOW_STALE_LIB; else export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1; fi
  echo "== $v"
  timeout -k 10 300 python3 tools/pmc_kernel.py $kern gpurun_out/pmcab_$v --config $cfg --steps 30 --warmup 20 --budget 1000 --spin-seconds 0 --no-cpu-baseline 2>&1 | grep -v "^{\|^}" | tr -d '\n' ; echo
done
