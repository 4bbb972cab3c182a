#!/bin/bash
# copies what tools/final_suite_r05.sh left under gpurun_out/final5/ into profiles/ (r05_final_*) and regenerates the summary
set -e
F=gpurun_out/final5; P=profiles
cp $F/bench.json $P/r05_final_bench.json; cp $F/bench_driver_shape.json $P/r05_final_bench_driver_shape.json
for b in 1 10 1000; do cp $F/bench_budget$b.json $P/r05_final_bench_budget$b.json; done
for c in cfg2 cfg2_one_group cfg3 cfg4; do cp $F/bench_${c}_under_rocprof.json $P/r05_final_bench_${c}_under_rocprof.json; cp $F/kt_$c/kt_kernel_stats.csv $P/r05_final_kernel_stats_$c.csv; done
cp $F/binding.txt $P/r05_final_binding_budget.txt; cp $F/binding_wide.txt $P/r05_final_binding_wide.txt; cp $F/general_model.txt $P/r05_final_general_model.txt
cp $F/ideal_step.txt $P/r05_final_ideal_step.txt; cp $F/ideal_profit.txt $P/r05_final_ideal_profit.txt; cp $F/closed_loop.txt $P/r05_final_closed_loop.txt
cp $F/sparse_floor.txt $P/r05_final_sparse_floor.txt; cp $F/keygen.txt $P/r05_final_keygen.txt
cp $F/vector_env.txt $P/r05_final_vector_env.txt; cp $F/small_env.txt $P/r05_final_small_env.txt
cp $F/binding_wide_float.txt $P/r05_final_binding_wide_float_models.txt
cp $F/kernel_stats_wide_binding.txt $P/r05_final_kernel_stats_wide_binding.txt
for b in 1000 10; do cp $F/kernel_stats_budget$b.csv $P/r05_final_kernel_stats_cfg2_budget$b.csv; done
(for f in $F/soak_*.txt; do echo "== $(basename $f .txt)"; tail -n 4 $f; done) > $P/r05_final_soak_parity.txt
python3 - <<'PY'
import json
out = {}
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    d = json.load(open(f"gpurun_out/final5/pmc/pmc_{c}.json"))
    out.update(d if c in d else {c: d})
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
python3 tools/summarize_r05.py
