#!/bin/bash
# Runs ON THE GPU BOX: parity of the budget-exact kernels + the binding-budget timings + kernel traces at budget 1000 / 10
export TMPDIR=/tmp
OUT=gpurun_out/bq
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "rest or click_walk or binding or exact or budget or median or soak or graph" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 120 python3 tools/soak_parity.py 60 > $OUT/soak.txt 2>&1 || { tail -20 $OUT/soak.txt; exit 1; }
tail -1 $OUT/soak.txt
ADCRAFT_REST_SPLIT=1 timeout -k 10 120 python3 tools/soak_parity.py 40 > $OUT/soak_pair.txt 2>&1 || { tail -20 $OUT/soak_pair.txt; exit 1; }
tail -1 $OUT/soak_pair.txt
timeout -k 10 200 python3 tools/exp_binding.py > $OUT/binding.txt 2>&1 || exit 1
head -4 $OUT/binding.txt
for b in 1000 10; do ADCRAFT_CLICK_WALK=0 bash tools/kt_budget.sh $b > $OUT/kt_$b.txt 2>&1 || exit 1; cat $OUT/kt_$b.txt; done
