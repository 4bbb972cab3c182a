#!/bin/bash
# Runs ON THE GPU BOX: long randomised GPU-vs-oracle soaks of the budget-exact kernels on the round's last build (~16 minutes)
OUT=gpurun_out/long_soak5; mkdir -p $OUT
ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 400 python3 tools/soak_parity.py 240 911 small > $OUT/at_once.txt 2>&1 || { tail -5 $OUT/at_once.txt; exit 1; }
tail -3 $OUT/at_once.txt
ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 400 python3 tools/soak_parity.py 240 912 > $OUT/rest_pair.txt 2>&1 || { tail -5 $OUT/rest_pair.txt; exit 1; }
tail -1 $OUT/rest_pair.txt
timeout -k 10 300 python3 tools/soak_parity.py 150 913 > $OUT/default_paths.txt 2>&1 || { tail -5 $OUT/default_paths.txt; exit 1; }
tail -1 $OUT/default_paths.txt
timeout -k 10 300 python3 tools/soak_parity.py 150 915 lists > $OUT/lists.txt 2>&1 || { tail -5 $OUT/lists.txt; exit 1; }
tail -3 $OUT/lists.txt
ADCRAFT_STREAM_GROUPS=4 ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 300 python3 tools/soak_parity.py 120 916 > $OUT/env_groups_rest_pair.txt 2>&1 || { tail -5 $OUT/env_groups_rest_pair.txt; exit 1; }
tail -1 $OUT/env_groups_rest_pair.txt
ADCRAFT_STREAM_GROUPS=2 timeout -k 10 300 python3 tools/soak_parity.py 100 917 explicit > $OUT/env_groups_explicit.txt 2>&1 || { tail -5 $OUT/env_groups_explicit.txt; exit 1; }
tail -1 $OUT/env_groups_explicit.txt
