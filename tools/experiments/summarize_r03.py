#!/usr/bin/env python3
"""profiles/r03_final_rocprof_summary.md from the files tools/final_suite_r03.sh produced (copied to profiles/r03_final_*)"""
import csv
import json


def top(path, n=6):
    return list(csv.DictReader(open(path)))[:n]


d2 = json.load(open("profiles/r03_final_bench_cfg2_under_rocprof.json"))
d3 = json.load(open("profiles/r03_final_bench_cfg3_under_rocprof.json"))
b = json.load(open("profiles/r03_final_bench.json"))
pm = json.load(open("profiles/pmc_traffic.json"))
L = ["# Round 3 — rocprofv3 evidence (1 x MI355X, builder-side gpurun box)\n",
     "All from `tools/final_suite_r03.sh`. Kernel traces: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-also --no-cpu-baseline`\n"
     "(the headline workload alone: 20 warm-up, 200 profiled and 200 timed steps) and `... bench.py --config cfg3 --no-cpu-baseline`; PMC: three separate\n"
     "`--pmc` passes per config (`tools/pmc_collect.py`: FETCH_SIZE, WRITE_SIZE, SQ counters), kernel-trace options only.\n"]
for name, path, bj in (("cfg2 (4096 x 256, dense; the bench line's workload)", "profiles/r03_final_kernel_stats_cfg2.csv", d2),
                       ("cfg3 (16384 x 1024, sparse)", "profiles/r03_final_kernel_stats_cfg3.csv", d3)):
    L.append(f"\n## {name}\n\n| kernel | calls | average (us) | share of GPU time |\n|---|---|---|---|")
    for r in top(path):
        L.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    L.append(f"\n`bench.py` in the same process (HIP events on the engine's stream): {bj['roofline']['kernel']} {bj['roofline']['kernel_ms'] * 1e3:.1f} us per launch, "
             f"{bj['ms_per_step'] * 1e3:.1f} us per step; `roofline.frac` {bj['roofline']['frac']:.4f}. (`k_ideal_profit` runs once, before the timed region: "
             "the episode-start ideal profit.)")
L.append("\n## PMC (per launch of the step kernel - `k_step_implicit_fast<false>` on cfg2/4/5, `k_step_implicit_sparse` on cfg3; mean over dispatches)\n\n| config | HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) | algorithmic bytes incl. "
         "metric-mode accumulators | VALU wave-instructions | lane-instructions per auction | LDS instructions | waves |\n|---|---|---|---|---|---|---|")
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    v = pm[c]
    L.append(f"| {c} | {v['hbm_bytes_per_launch'] / 1e6:.1f} MB | {v['algorithmic_bytes_incl_metric_mode'] / 1e6:.1f} MB | {v['valu']['wave_instructions_per_launch']:.3g} | "
             f"{v['valu']['valu_lane_instructions_per_auction']:.1f} | {v['valu']['lds_instructions']:.3g} | {v['valu']['waves']:.0f} |")
L.append("\nTraffic is within 2 % of the algorithmic bytes on every config: nothing is re-read. The dense kernel issues 56.5 VALU lane-instructions per\n"
         "auction (unchanged from round 2); the sparse kernel 109.5 per auction = 2.30e8 wave-instructions per launch (round 2: 2.73e8), of which the\n"
         "auctions themselves are about a quarter - the rest is the per-keyword work (parameters, brackets, volume, outputs) that cfg3's 8-auction\n"
         "keywords cannot amortise. `roofline_valu.frac` in the bench line prices them against one wave-instruction per SIMD per 2 cycles.\n")
ds = json.load(open("profiles/r03_final_bench_driver_shape.json"))
L.append("## The driver's command in the same run\n")
L.append(f"`python bench.py --steps 20 --warmup 5` (the shape the driver uses): value {ds['value']:.4g} keyword-steps/s, {ds['ms_per_step']:.4f} ms/step, "
         f"kernel {ds['roofline']['kernel_ms'] * 1e3:.1f} us, event records inside the timed region: {ds.get('event_records_in_timed_region')}; also: "
         + ", ".join(f"{k} {v['ms_per_step']:.3f} ms/step (HBM frac {v['roofline']['frac']:.3f})" for k, v in ds['also'].items()) + ".\n")
L.append("`python bench.py` (defaults: 200 steps):\n")
L.append(f"value {b['value']:.4g} keyword-steps/s, {b['ms_per_step']:.4f} ms/step; also: "
         + ", ".join(f"{k} {v['ms_per_step']:.3f} ms/step (HBM frac {v['roofline']['frac']:.3f})" for k, v in b['also'].items())
         + f"; host_step {b['host_step']['ms']} ms (uint16 counts {b['host_step']['ms_u16_counts']} ms).\n")
L.append("Other files: `r03_final_binding_budget.txt` (binding budgets, single env, EXPLICIT), `r03_final_binding_wide.txt` (K = 512 / 1024),\n"
         "`r03_final_general_model.txt` (the default ImplicitKeyword), `r03_final_bench_budget*.json`, `r03_final_vector_env.txt`, `r03_final_small_env.txt`,\n"
         "`r03_final_ideal_step.txt` (per-step ideal profit: contender lists vs the full grid scan), `r03_final_closed_loop.txt`, `r03_final_sparse_floor.txt`\n"
         "(the sparse kernel with the auctions / the dead keywords removed: where its time goes), `r03_final_soak_parity.txt` (randomised GPU-vs-oracle\n"
         "steps, all three models and the sparse kernel forced onto every shape, bit-exact), `r03_stream_battery.txt` (Philox4x32-7 on the production\n"
         "counter layout), `r02_issue_rates.md` (the per-instruction cost model, unchanged).\n")
open("profiles/r03_final_rocprof_summary.md", "w").write("\n".join(L))
print("written")
