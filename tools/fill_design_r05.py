#!/usr/bin/env python3
"""fills DESIGN.md section 6's placeholders (BENCH_*, PMC_*, KT_*, ...) from profiles/r05_final_* (after tools/copy_final_r05.sh)"""
import csv
import json
import re


def last_json_line(path):
    return json.loads([ln for ln in open(path) if ln.startswith("{")][-1])


def sci(x):
    m, e = f"{x:.2e}".split("e")
    return f"{m}·10{str(int(e)).translate(str.maketrans('0123456789', '⁰¹²³⁴⁵⁶⁷⁸⁹'))}"


P = "profiles/r05_final_"
b = last_json_line(P + "bench.json")
ds = last_json_line(P + "bench_driver_shape.json")
pm = json.load(open("profiles/pmc_traffic.json"))
vol = {"cfg2": 128.0, "cfg3": 16.0 * 0.5, "cfg4": 128.0, "cfg5": 128.0}
rep = {}
for c in ("cfg2", "cfg3", "cfg4", "cfg5"):
    r = b if c == "cfg2" else b["also"][c]
    C = c.upper()
    rep[f"BENCH_{C}_ONE"] = f"{r['env_groups']['ms_per_step_as_one_group']:.3f}"
    rep[f"BENCH_{C}_VALU"] = (f"{r['roofline_valu']['frac']:.2f} / {r['roofline_valu']['frac_4cycle']:.2f}" if r.get("roofline_valu", {}).get("frac") else "-")
    rep[f"BENCH_{C}_VAL"] = sci(r["value"])
    rep[f"BENCH_{C}_AUC"] = sci(r["value"] * vol[c])
    rep[f"BENCH_{C}_K"] = f"{r['roofline']['kernel_ms']:.3f}"
    rep[f"BENCH_{C}_FRAC"] = f"{r['roofline']['frac']:.3f}"
    rep[f"BENCH_{C}"] = f"{r['ms_per_step']:.4f}" if c == "cfg2" else f"{r['ms_per_step']:.3f}"
rep["BENCH_CFG2_DRV"] = f"{ds['ms_per_step']:.4f}"
rep["PMC_LINE"] = "; ".join(
    f"{c} {pm[c]['hbm_bytes_per_launch'] / 1e6:.1f} MB against {pm[c]['algorithmic_bytes_incl_metric_mode'] / 1e6:.1f} MB algorithmic incl. the metric word | "
    f"{pm[c]['valu']['wave_instructions_per_launch']:.3e} ({pm[c]['valu']['valu_lane_instructions_per_auction']:.1f} lane-instructions per auction)" for c in ("cfg2", "cfg3", "cfg4", "cfg5"))
rep["PMC_HASH"] = pm["cfg2"]["library_source_hash"]


def kt(c, name):
    for r in csv.DictReader(open(f"{P}kernel_stats_{c}.csv")):
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return float("nan")


rep["KT_CFG2_ONE"] = f"{kt('cfg2_one_group', 'k_step_implicit_fast'):.1f}"
rep["KT_CFG2_GRP"] = f"{kt('cfg2', 'k_step_implicit_fast'):.1f}"
rep["KT_CFG3"] = f"{kt('cfg3', 'k_step_implicit_sparse'):.1f}"
rep["KT_CFG4"] = f"{kt('cfg4', 'k_step_implicit_fast'):.1f}"
rep["IDEAL_K"] = " / ".join(f"{kt(c, 'k_ideal_profit') / 1e3:.2f}" for c in ("cfg2", "cfg3", "cfg4"))
rows = list(csv.DictReader(open(P + "kernel_stats_cfg2_budget1000.csv")))[:3]
rep["KSTATS_B1000"] = ", ".join(f"`{re.sub(r'^void adck::|^adck::', '', r['Name']).split('(')[0]}` {float(r['AverageNs']) / 1e3:.0f} µs" for r in rows)
cl = last_json_line(P + "closed_loop.txt")
rep["CLOSED_LOOP_VAL"] = sci(cl["keyword_steps_per_s"])
rep["CLOSED_LOOP"] = f"{cl['ms_per_loop_step']:.3f}"
coll = b["also"]["cfg4"]["collective"]["ms_per_call"]
rep["COLLECTIVE_LINE"] = f"own reduction kernels {coll['own_reduction_kernels']:.3f} ms + `ncclAllReduce` of 3080 doubles {coll['allreduce'] * 1e3:.1f} µs"
cb = b["cpu_baseline"]
rep["CPU_LINE"] = f"{sci(cb['single_thread']['value'])} / {sci(cb['value'])}"
tot_steps = tot_runs = 0
for ln in open(P + "soak_parity.txt"):
    m = re.search(r"soak ok: (\d+) engines, (\d+) steps", ln)
    if m:
        tot_runs += 1
        tot_steps += int(m.group(2))
long_steps = 0
try:
    for ln in open("profiles/r05_long_soak.txt"):
        m = re.search(r"soak ok: (\d+) engines, (\d+) steps", ln)
        if m:
            long_steps += int(m.group(2))
except FileNotFoundError:
    pass
rep["SOAK_LINE"] = (f"{tot_runs} modes, {tot_steps} randomised steps against the oracle (all three models, the sparse kernel forced, both GENERAL passes, the rest-of-day pair, "
                    f"at-once parking, click lists, three env groups forced) + the long soak of the budget-exact kernels and of forced env groups: {long_steps} steps (`profiles/r05_long_soak.txt`)")
s = open("DESIGN.md").read()
for k in sorted(rep, key=len, reverse=True):
    s = s.replace(k, rep[k])
open("DESIGN.md", "w").write(s)
left = sorted(set(re.findall(r"\b(?:BENCH|PMC|KT|KSTATS|IDEAL_K|CLOSED_LOOP|COLLECTIVE_LINE|CPU_LINE|SOAK_LINE)[A-Z0-9_]*\b", s)))
print("filled", len(rep), "left:", left)
