#!/usr/bin/env python3
"""profiles/r05_final_rocprof_summary.md from the files tools/final_suite_r05.sh produced (copied to profiles/r05_final_* by
tools/copy_final_r05.sh)"""
import csv
import json


def last_json_line(path):
    return json.loads([ln for ln in open(path) if ln.startswith("{")][-1])


def top(path, n=6):
    return list(csv.DictReader(open(path)))[:n]


P = "profiles/r05_final_"
runs = {c: last_json_line(f"{P}bench_{c}_under_rocprof.json") for c in ("cfg2", "cfg2_one_group", "cfg3", "cfg4")}
b = last_json_line(P + "bench.json")
ds = last_json_line(P + "bench_driver_shape.json")
pm = json.load(open("profiles/pmc_traffic.json"))
L = ["# Round 5 - rocprofv3 evidence (1 x MI355X, builder-side gpurun box)\n",
     "All from `tools/final_suite_r05.sh`.  Kernel traces: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-also --no-cpu-baseline` (the headline\n"
     "workload alone), `... bench.py --config cfg3 --no-cpu-baseline`, `... --config cfg4` (the N > 1 workload's per-GPU shard, with its collective on a\n"
     "one-rank communicator); PMC: three separate `--pmc` passes per config (`tools/pmc_collect.py`: FETCH_SIZE, WRITE_SIZE, SQ counters), kernel-trace\n"
     "options only, one env group (every dispatch the full batch).\n",
     "\n**Env groups.**  From 2048 envs on the engine launches a CHAIN of device-resident steps as four env groups on four streams (two for sparse\n"
     "keyword sets; `profiles/r05_stream_groups.txt`): in a trace of the default run the step kernel therefore appears four times per step at a quarter\n"
     "of the batch (plus the full-batch launches of `bench.py`'s profiled and one-group passes and of the first step behind every barrier), and its\n"
     "average duration there is that of a launch which SHARES the chip with three others.  `k_spin_ticks` (51 calls) is the one-time probe that picks\n"
     "the groups' streams, one per hardware queue.  The table that `roofline.kernel_ms` is to be checked against is the second one, the same command under\n"
     "`ADCRAFT_STREAM_GROUPS=1`: every launch the full batch, alone on the chip.\n"]
for name, c in (("cfg2 (4096 x 256, dense; the bench line's workload) - as the engine schedules it (4 env groups)", "cfg2"),
                ("cfg2, the same command under ADCRAFT_STREAM_GROUPS=1 (one group: full-batch launches)", "cfg2_one_group"),
                ("cfg3 (16384 x 1024, sparse; 2 env groups)", "cfg3"),
                ("cfg4 (8192 x 1024: one GPU's shard of 65536 x 1024)", "cfg4")):
    bj = runs[c]
    L.append(f"\n## {name}\n\n| kernel | calls | average (us) | share of GPU time |\n|---|---|---|---|")
    for r in top(f"{P}kernel_stats_{c}.csv"):
        L.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    L.append(f"\n`bench.py` in the same process (HIP events on the engine's stream): {bj['roofline']['kernel']} {bj['roofline']['kernel_ms'] * 1e3:.1f} us per launch, "
             f"{bj['ms_per_step'] * 1e3:.1f} us per step by the host clock, {bj['timed_region']['gpu_ms_per_step'] * 1e3:.1f} us by one event pair around the timed region; "
             f"`roofline.frac` {bj['roofline']['frac']:.4f}; env groups in the timed region: {bj['env_groups']['timed_region']} "
             f"(the same steps as one group: {bj['env_groups']['ms_per_step_as_one_group'] * 1e3:.1f} us per step).")
    if "collective" in bj:
        L.append(f"The collective: {json.dumps(bj['collective']['ms_per_call'])} ms per call, {bj['collective']['calls_in_timed_region']} call(s), {bj['collective']['ranks']} rank(s).")
L.append("\n## PMC (per launch of the step kernel - `k_step_implicit_fast<false>` on cfg2/4/5, `k_step_implicit_sparse` on cfg3; mean over dispatches)\n\n"
         "| config | HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) | algorithmic bytes incl. metric-mode accumulators | VALU wave-instructions | lane-instructions per auction | LDS instructions | waves |\n"
         "|---|---|---|---|---|---|---|")
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    v = pm[c]
    L.append(f"| {c} | {v['hbm_bytes_per_launch'] / 1e6:.1f} MB | {v['algorithmic_bytes_incl_metric_mode'] / 1e6:.1f} MB | {v['valu']['wave_instructions_per_launch']:.3g} | "
             f"{v['valu']['valu_lane_instructions_per_auction']:.1f} | {v['valu']['lds_instructions']:.3g} | {v['valu']['waves']:.0f} |")
L.append("\nThe metric-mode accumulators are a 32-bit word per keyword (+8 B per keyword-step).  `roofline_valu` in the bench line prices the VALU\n"
         "instructions against one wave-instruction per SIMD per 2 cycles (`frac`) and per 4 cycles (`frac_4cycle`: what this instruction mix can\n"
         "reach, `profiles/r02_issue_rates.md`); the counters are quoted only for the library build they were taken on (`library_source_hash`).\n")
L.append("## The driver's command in the same run\n")
L.append(f"`python bench.py --steps 20 --warmup 5` (the shape the driver uses): value {ds['value']:.4g} keyword-steps/s, {ds['ms_per_step']:.4f} ms/step "
         f"({ds['clock_spin_steps']} untimed clock-spin steps before it), kernel {ds['roofline']['kernel_ms'] * 1e3:.1f} us; also: "
         + ", ".join(f"{k} {v['ms_per_step']:.3f} ms/step (HBM frac {v['roofline']['frac']:.3f})" for k, v in ds['also'].items()) + ".\n")
L.append("`python bench.py` (defaults: 200 steps):\n")
L.append(f"value {b['value']:.4g} keyword-steps/s, {b['ms_per_step']:.4f} ms/step; also: "
         + ", ".join(f"{k} {v['ms_per_step']:.3f} ms/step (HBM frac {v['roofline']['frac']:.3f})" for k, v in b['also'].items())
         + f"; host_step {b['host_step']['ms']} ms (uint16 counts {b['host_step']['ms_u16_counts']} ms).\n")
L.append("Other files of the suite: `r05_final_binding_budget.txt`, `r05_final_binding_wide.txt`, `r05_final_binding_wide_float_models.txt`, `r05_final_bench_budget*.json`,\n"
         "`r05_final_kernel_stats_cfg2_budget*.csv` (binding budgets), `r05_final_general_model.txt` (the default ImplicitKeyword), `r05_final_ideal_step.txt`,\n"
         "`r05_final_ideal_profit.txt`, `r05_final_closed_loop.txt`, `r05_final_sparse_floor.txt`, `r05_final_keygen.txt`, `r05_final_vector_env.txt`,\n"
         "`r05_final_small_env.txt`, `r05_final_soak_parity.txt` (randomised GPU-vs-oracle steps: all three models, the sparse kernel forced onto every shape,\n"
         "both GENERAL passes, the rest-of-day pair, the click lists, three env groups forced - bit-exact), `r05_stream_groups.txt` (env groups: every shape measured).\n")
open("profiles/r05_final_rocprof_summary.md", "w").write("\n".join(L))
print("written")
