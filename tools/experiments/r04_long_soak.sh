#!/bin/bash
# Runs ON THE GPU BOX: long randomised GPU-vs-oracle soaks of the budget-exact kernels as they stand at the end of round 4
OUT=gpurun_out/long_soak; mkdir -p $OUT
ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 400 python3 tools/soak_parity.py 240 901 small > $OUT/at_once.txt 2>&1 || { tail -5 $OUT/at_once.txt; exit 1; }
tail -3 $OUT/at_once.txt
ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 400 python3 tools/soak_parity.py 240 902 > $OUT/rest_pair.txt 2>&1 || { tail -5 $OUT/rest_pair.txt; exit 1; }
tail -1 $OUT/rest_pair.txt
ADCRAFT_REST_SPLIT=1 ADCRAFT_ROWS_WIDE=1 timeout -k 10 300 python3 tools/soak_parity.py 150 903 > $OUT/rows_lane_per_keyword.txt 2>&1 || { tail -5 $OUT/rows_lane_per_keyword.txt; exit 1; }
tail -1 $OUT/rows_lane_per_keyword.txt
ADCRAFT_ROWS_WIDE=2 timeout -k 10 300 python3 tools/soak_parity.py 150 904 > $OUT/rows_512.txt 2>&1 || { tail -5 $OUT/rows_512.txt; exit 1; }
tail -1 $OUT/rows_512.txt
timeout -k 10 300 python3 tools/soak_parity.py 150 905 lists > $OUT/lists.txt 2>&1 || { tail -5 $OUT/lists.txt; exit 1; }
tail -3 $OUT/lists.txt
