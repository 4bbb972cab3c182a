#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of tools/profile_shape.py N K budget -> gpurun_out/kts_N_K_budget/
export TMPDIR=/tmp
OUT=gpurun_out/kts_$1_$2_$3; rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 tools/profile_shape.py $1 $2 $3 > $OUT.log 2> $OUT.err || exit 1
rm -f $OUT/*kernel_trace.csv
echo "== $1 x $2 budget $3"; python3 - $OUT/kt_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
