#!/bin/bash
# Runs ON THE GPU BOX: bench one config under a sweep of one scheduling override.  usage: r03_sweep.sh <tag> <cfg> <ENVVAR> v1 v2 ...
TAG=$1; CFG=$2; VAR=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for val in "$@"; do
  env $VAR=$val python bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_${CFG}_$val.json 2> $OUT/bench_${CFG}_$val.err
  python tools/bench_summary.py "$VAR=$val" < $OUT/bench_${CFG}_$val.json
done
