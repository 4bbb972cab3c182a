#!/bin/bash
# GPU box: phase timers only.  usage: r03_timing.sh <tag> <cfg> [ENV=val ...]
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
env "$@" ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/timing.so ADCRAFT_ALLOW_STALE_LIB=1 python tools/exp_fast_timing.py $CFG 2>&1 | tee -a $OUT/timing_$CFG.txt
