#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/final_suite_r05.sh): the round's final numbers into gpurun_out/final5/.
# Copy what is to be judged into profiles/ afterwards (tools/copy_final_r05.sh: r05_final_*).
# In parts (a gpurun call is limited to 20 minutes): bash tools/final_suite_r05.sh bench | timings | soaks1 | soaks2
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/final5
mkdir -p $OUT
PART=${1:-bench}
if [ $PART = bench ]; then
# the PMC passes first (the bench lines print them only from a record with this build's source hash), then
for c in cfg2 cfg3 cfg4 cfg5; do timeout -k 10 400 python3 tools/pmc_collect.py $c $OUT/pmc > $OUT/pmc_$c.log 2>&1 || exit 1; done
python3 - <<'PY' || exit 1
import json
out = {}
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    d = json.load(open(f"gpurun_out/final5/pmc/pmc_{c}.json"))
    out.update(d if c in d else {c: d})
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)     # (on the box: the bench lines below then carry this build's counters)
PY
echo "pmc done"
# the driver's command (its shape: --steps 20 --warmup 5), the default invocation, then the same workloads alone under the kernel trace
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench.err || exit 1
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2>> $OUT/bench.err || exit 1
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg2 -o kt -- python3 bench.py --no-also --no-cpu-baseline > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/kt_cfg2.err || exit 1
ADCRAFT_STREAM_GROUPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg2_one_group -o kt -- python3 bench.py --no-also --no-cpu-baseline > $OUT/bench_cfg2_one_group_under_rocprof.json 2> $OUT/kt_cfg2_one_group.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg3 -o kt -- python3 bench.py --config cfg3 --no-cpu-baseline > $OUT/bench_cfg3_under_rocprof.json 2> $OUT/kt_cfg3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg4 -o kt -- python3 bench.py --config cfg4 --no-cpu-baseline --steps 120 > $OUT/bench_cfg4_under_rocprof.json 2> $OUT/kt_cfg4.err || exit 1
echo "kernel traces done"

rm -rf $OUT/pmc/*_fetch $OUT/pmc/*_write $OUT/pmc/*_sq $OUT/kt_cfg2/*kernel_trace.csv $OUT/kt_cfg2_one_group/*kernel_trace.csv $OUT/kt_cfg3/*kernel_trace.csv $OUT/kt_cfg4/*kernel_trace.csv
fi
if [ $PART = timings ]; then
for b in 1000 10 1; do timeout -k 10 200 python3 bench.py --config cfg2 --budget $b --no-cpu-baseline > $OUT/bench_budget$b.json 2>> $OUT/bench.err || exit 1; done
timeout -k 10 200 python3 tools/exp_binding.py > $OUT/binding.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/exp_binding_wide.py > $OUT/binding_wide.txt 2>&1 || exit 1
timeout -k 10 400 python3 tools/exp_binding_wide_float.py > $OUT/binding_wide_float.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_general.py > $OUT/general_model.txt 2>&1 || exit 1
timeout -k 10 100 python3 tools/profile_general_small.py >> $OUT/general_model.txt 2>&1 || exit 1
timeout -k 10 100 python3 tools/profile_general_small.py 1000 >> $OUT/general_model.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_ideal_step.py > $OUT/ideal_step.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_ideal_profit.py > $OUT/ideal_profit.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_closed_loop.py > $OUT/closed_loop.txt 2>&1 || exit 1
ADCRAFT_FAST_VARIANT=2 timeout -k 10 200 python3 tools/exp_sparse_floor.py 40 > $OUT/sparse_floor.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_keygen.py > $OUT/keygen.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_vector_env.py > $OUT/vector_env.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/measure_small_env.py > $OUT/small_env.txt 2>&1 || exit 1
# (kernel traces of binding steps as ONE env group: a launch's duration is then its own, not that of a launch sharing the chip with three others)
for shape in "2048 1024 4000" "2048 1024 40" "4096 512 4000" "4096 512 40"; do ADCRAFT_STREAM_GROUPS=1 bash tools/kt_shape.sh $shape >> $OUT/kernel_stats_wide_binding.txt 2>&1 || exit 1; done
for b in 1000 10; do ADCRAFT_STREAM_GROUPS=1 ADCRAFT_CLICK_WALK=0 bash tools/kt_budget.sh $b > /dev/null 2>&1 || exit 1; cp gpurun_out/kt_b$b/kt_kernel_stats.csv $OUT/kernel_stats_budget$b.csv; done
echo "timings done"
fi
if [ $PART = soaks1 ]; then
timeout -k 10 300 python3 tools/soak_parity.py 100 401 > $OUT/soak_implicit.txt 2>&1 || exit 1
ADCRAFT_FAST_VARIANT=2 ADCRAFT_FAST_TILE_KW=256 timeout -k 10 300 python3 tools/soak_parity.py 80 404 > $OUT/soak_implicit_sparse_kernel.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/soak_parity.py 80 402 explicit > $OUT/soak_explicit.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/soak_parity.py 80 403 general > $OUT/soak_general.txt 2>&1 || exit 1
for f in $OUT/soak_*.txt; do echo "$f: $(tail -n1 $f)"; done
fi
if [ $PART = soaks2 ]; then
ADCRAFT_GENERAL_SMALL=0 timeout -k 10 300 python3 tools/soak_parity.py 60 408 general > $OUT/soak_general_lane_per_keyword.txt 2>&1 || exit 1
ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 300 python3 tools/soak_parity.py 60 405 > $OUT/soak_implicit_rest_pair.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/soak_parity.py 60 409 lists > $OUT/soak_implicit_lists.txt 2>&1 || exit 1
ADCRAFT_REST_SPLIT=1 ADCRAFT_CLICK_WALK=0 timeout -k 10 300 python3 tools/soak_parity.py 60 410 small > $OUT/soak_implicit_at_once.txt 2>&1 || exit 1
ADCRAFT_STREAM_GROUPS=3 timeout -k 10 300 python3 tools/soak_parity.py 50 411 > $OUT/soak_implicit_env_groups.txt 2>&1 || exit 1
for f in $OUT/soak_*.txt; do echo "$f: $(tail -n1 $f)"; done
fi
echo "part $PART done"
