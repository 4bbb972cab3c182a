#!/bin/bash
# Runs ON THE GPU BOX: parity + binding timings of the wide shapes (K = 512 / 1024)
export TMPDIR=/tmp
OUT=gpurun_out/wq; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "rest or click_walk or binding or exact or budget or at_once or wide or large" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
ADCRAFT_REST_SPLIT=1 timeout -k 10 120 python3 tools/soak_parity.py 50 > $OUT/soak_pair.txt 2>&1 || { tail -20 $OUT/soak_pair.txt; exit 1; }
tail -1 $OUT/soak_pair.txt
timeout -k 10 200 python3 tools/exp_binding_wide.py 2>&1 | tee $OUT/binding_wide.txt || exit 1
bash tools/kt_shape.sh 2048 1024 4000 && bash tools/kt_shape.sh 2048 1024 40
