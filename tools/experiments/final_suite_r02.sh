#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/final_suite_r02.sh): the round's final numbers into gpurun_out/final2/.
# Copy what is to be judged into profiles/ afterwards (r02_final_*).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/final2
rm -rf $OUT; mkdir -p $OUT
# the driver's command, then the same workload alone under the kernel trace (its k_step_implicit_fast average must agree with roofline.kernel_ms)
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg2 -o kt -- python3 bench.py --no-also --no-cpu-baseline > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/kt_cfg2.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg3 -o kt -- python3 bench.py --config cfg3 --no-cpu-baseline > $OUT/bench_cfg3_under_rocprof.json 2> $OUT/kt_cfg3.err || exit 1
echo "kernel traces done"
for c in cfg2 cfg3; do timeout -k 10 300 python3 tools/pmc_collect.py $c $OUT/pmc > $OUT/pmc_$c.log 2>&1 || exit 1; done
rm -rf $OUT/pmc/*_fetch $OUT/pmc/*_write $OUT/pmc/*_sq $OUT/kt_cfg2/*kernel_trace.csv $OUT/kt_cfg3/*kernel_trace.csv
echo "pmc done"
for b in 1000 10 1; do timeout -k 10 200 python3 bench.py --config cfg2 --budget $b --no-cpu-baseline > $OUT/bench_budget$b.json 2>> $OUT/bench.err || exit 1; done
timeout -k 10 200 python3 tools/exp_binding.py > $OUT/binding.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_vector_env.py > $OUT/vector_env.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_host_pieces.py > $OUT/host_pieces.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_small_env.py > $OUT/small_env.txt 2>&1 || exit 1
echo "timings done"
timeout -k 10 300 python3 tools/soak_parity.py 150 101 > $OUT/soak_implicit.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/soak_parity.py 150 102 explicit > $OUT/soak_explicit.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/soak_parity.py 150 103 general > $OUT/soak_general.txt 2>&1 || exit 1
tail -n1 $OUT/soak_implicit.txt; tail -n1 $OUT/soak_explicit.txt; tail -n1 $OUT/soak_general.txt
echo done
