#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the cfg2 bench at a binding budget with a variant build.  usage: r04_kt_variant.sh <variant|product> <budget>
export TMPDIR=/tmp
V=$1; B=$2
if [ $V != product ]; then export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$V.so ADCRAFT_ALLOW_STALE_LIB=1; fi
export ADCRAFT_CLICK_WALK=0
OUT=gpurun_out/ktv_${V}_$B; rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 bench.py --config cfg2 --budget $B --no-also --no-cpu-baseline --steps 60 --warmup 10 > $OUT.json 2> $OUT.err || exit 1
rm -f $OUT/*kernel_trace.csv
echo "== $V budget $B"; head -5 $OUT/kt_kernel_stats.csv | cut -d, -f1,2,4,6,7 | cut -c1-150
