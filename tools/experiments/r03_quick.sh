#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/r03_quick.sh <tag> [cfg]): sparse-kernel parity subset, one bench config, the phase timers.
TAG=${1:-q}; CFG=${2:-cfg3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse or brackets or cfg3 or fast_pass or non_finite or soak" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
python bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
python tools/bench_summary.py $CFG < $OUT/bench_$CFG.json
if [ -f adcraft_amd/lib/variants/timing.so ]; then
  ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/timing.so ADCRAFT_ALLOW_STALE_LIB=1 python tools/exp_fast_timing.py $CFG > $OUT/timing_$CFG.txt 2>&1; cat $OUT/timing_$CFG.txt
fi
