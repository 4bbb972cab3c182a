#!/bin/bash
# Runs ON THE GPU BOX: the round-5 iteration loop - GPU parity tests, then the keyword-parallel kernels' times on cfg2 / cfg3 for the
# product build and for any variant builds given (adcraft_amd/lib/variants/<name>.so), then a short bench line.
# usage: tools/r05_quick.sh [variant ...]        (output: gpurun_out/r05q/)
export TMPDIR=/tmp
OUT=gpurun_out/r05q; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
tail -4 $OUT/tests.log
[ $rc = 0 ] || exit $rc
for v in product "$@"; do
  if [ $v = product ]; then unset ADCRAFT_HIP_LIB ADCRAFT_ALLOW_STALE_LIB; else export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1; fi
  for cfg in cfg3 cfg2; do
    timeout -k 10 120 python3 tools/exp_sparse_ablation.py $cfg 60 1 2>&1 | tail -1 | tee -a $OUT/kernels.txt || exit 1
  done
done
unset ADCRAFT_HIP_LIB ADCRAFT_ALLOW_STALE_LIB
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 tools/bench_summary.py $OUT/bench.json
