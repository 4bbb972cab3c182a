#!/bin/bash
# copies what tools/final_suite_r03.sh left under gpurun_out/final3/ into profiles/ (r03_final_*) and regenerates the summary
set -e
F=gpurun_out/final3; P=profiles
cp $F/bench.json $P/r03_final_bench.json; cp $F/bench_driver_shape.json $P/r03_final_bench_driver_shape.json
for b in 1 10 1000; do cp $F/bench_budget$b.json $P/r03_final_bench_budget$b.json; done
cp $F/bench_cfg2_under_rocprof.json $P/r03_final_bench_cfg2_under_rocprof.json; cp $F/bench_cfg3_under_rocprof.json $P/r03_final_bench_cfg3_under_rocprof.json
cp $F/kt_cfg2/kt_kernel_stats.csv $P/r03_final_kernel_stats_cfg2.csv; cp $F/kt_cfg3/kt_kernel_stats.csv $P/r03_final_kernel_stats_cfg3.csv
cp $F/binding.txt $P/r03_final_binding_budget.txt; cp $F/binding_wide.txt $P/r03_final_binding_wide.txt; cp $F/general_model.txt $P/r03_final_general_model.txt
cp $F/ideal_step.txt $P/r03_final_ideal_step.txt; cp $F/closed_loop.txt $P/r03_final_closed_loop.txt; cp $F/sparse_floor.txt $P/r03_final_sparse_floor.txt
cp $F/vector_env.txt $P/r03_final_vector_env.txt; cp $F/small_env.txt $P/r03_final_small_env.txt
cp $F/binding_ctr.txt $P/r03_final_binding_click_walk.txt; cp $F/binding_wide_float.txt $P/r03_final_binding_wide_float_models.txt
for b in 1000 10; do cp $F/kernel_stats_budget$b.csv $P/r03_final_kernel_stats_cfg2_budget$b.csv; done
(for f in soak_implicit soak_implicit_sparse_kernel soak_implicit_rest_pair soak_explicit soak_general; do echo "== $f"; tail -n 6 $F/$f.txt; done) > $P/r03_final_soak_parity.txt
python3 - <<'PY'
import json
out = {}
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    d = json.load(open(f"gpurun_out/final3/pmc/pmc_{c}.json"))
    out.update(d if c in d else {c: d})
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
python3 tools/summarize_r03.py
