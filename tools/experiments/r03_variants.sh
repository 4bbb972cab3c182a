#!/bin/bash
# GPU box: bench one config with each library variant under adcraft_amd/lib/variants (plus the default build).
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_${CFG}_default.json 2> $OUT/bench_${CFG}_default.err; python tools/bench_summary.py default < $OUT/bench_${CFG}_default.json
for v in "$@"; do
  ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1 python bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_${CFG}_$v.json 2> $OUT/bench_${CFG}_$v.err
  python tools/bench_summary.py $v < $OUT/bench_${CFG}_$v.json
done
