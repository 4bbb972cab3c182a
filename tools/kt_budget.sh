#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of bench.py on cfg2 at a binding budget ($1) -> gpurun_out/kt_b$1/kt_kernel_stats.csv
export TMPDIR=/tmp
B=$1
OUT=gpurun_out/kt_b$B
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 bench.py --config cfg2 --budget $B --no-also --no-cpu-baseline --steps 60 --warmup 10 > $OUT.json 2> $OUT.err || exit 1
rm -f $OUT/*kernel_trace.csv
head -6 $OUT/kt_kernel_stats.csv | cut -c1-160
