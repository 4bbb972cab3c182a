#!/usr/bin/env python3
"""regenerates the measured rows of DESIGN.md section 6 (the four bench rows, the PMC / kernel-trace paragraph, the ideal-profit,
closed-loop, collective, CPU and soak rows) from profiles/r05_final_* and profiles/pmc_traffic.json (after tools/copy_final_r05.sh);
idempotent: rows are found by their first cell"""
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


def row(c, label, bold_frac=False, note=""):
    r = b if c == "cfg2" else b["also"][c]
    g = r["env_groups"]["timed_region"]
    one = f" ({r['env_groups']['ms_per_step_as_one_group']:.3f}" + (f"; driver shape, 20 steps: {ds['ms_per_step']:.4f})" if c == "cfg2" else ")") if g > 1 else ""
    ms = f"**{r['ms_per_step']:.4f}**" if c == "cfg2" else f"{r['ms_per_step']:.3f}"
    val = f"**{sci(r['value'])}**" if c == "cfg2" else sci(r["value"])
    frac = f"{r['roofline']['frac']:.3f}"
    rv = r.get("roofline_valu", {})
    valu = f"{rv['frac']:.2f} / {rv['frac_4cycle']:.2f}" if rv.get("frac") else "-"
    return (f"| {label}, {g} group{'s' if g > 1 else ''} | {ms}{one} | {val} | {sci(r['value'] * vol[c])} | {r['roofline']['kernel_ms']:.3f} ms | "
            f"{'**' + frac + '**' if bold_frac else frac} | {valu}{note} |")


def kt(c, name):
    for r in csv.DictReader(open(f"{P}kernel_stats_{c}.csv")):
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return float("nan")


s = open("DESIGN.md").read()


def replace_row(first_cell_start, new_line):
    global s
    m = re.search(r"^\| " + re.escape(first_cell_start) + r".*$", s, flags=re.M)
    assert m, first_cell_start
    s = s[:m.start()] + new_line + s[m.end():]


replace_row("**cfg2** 4096 × 256 dense (the bench line)", row("cfg2", "**cfg2** 4096 × 256 dense (the bench line)"))
replace_row("cfg3 16384 × 1024 sparse", row("cfg3", "cfg3 16384 × 1024 sparse", bold_frac=True))
replace_row("cfg4 8192 × 1024 (one GPU's shard of 65536 × 1024), with its collective", row("cfg4", "cfg4 8192 × 1024 (one GPU's shard of 65536 × 1024), with its collective", note=" (part of the mix issues in 2 cycles)"))
replace_row("cfg5 2048 × 1024 + drift (shard of 16384 × 1024)", row("cfg5", "cfg5 2048 × 1024 + drift (shard of 16384 × 1024)"))
pmc_line = "; ".join(
    f"{c} {pm[c]['hbm_bytes_per_launch'] / 1e6:.1f} MB against {pm[c]['algorithmic_bytes_incl_metric_mode'] / 1e6:.1f} MB algorithmic incl. the metric word | "
    f"{pm[c]['valu']['wave_instructions_per_launch']:.3e} ({pm[c]['valu']['valu_lane_instructions_per_auction']:.1f} lane-instructions per auction)" for c in ("cfg2", "cfg3", "cfg4", "cfg5"))
para = (f"PMC per launch, one group (2 × FETCH_SIZE + WRITE_SIZE | VALU wave-instructions): {pmc_line}\n"
        f"(`profiles/pmc_traffic.json`, library sources `{pm['cfg2']['library_source_hash']}`). `rocprofv3 --kernel-trace --stats` of the cfg2 command as one group:\n"
        f"`k_step_implicit_fast<false>` {kt('cfg2_one_group', 'k_step_implicit_fast'):.1f} µs; as the engine schedules it the same kernel appears four times per step at a quarter of the batch\n"
        f"({kt('cfg2', 'k_step_implicit_fast'):.1f} µs on average, sharing the chip with the other groups' launches); `k_step_implicit_sparse` {kt('cfg3', 'k_step_implicit_sparse'):.1f} µs (cfg3, 2 groups),\n"
        f"`k_step_implicit_fast<false>` {kt('cfg4', 'k_step_implicit_fast'):.1f} µs (cfg4, 1 group).")
m = re.search(r"^PMC per launch, one group .*?\(cfg4, 1 group\)\.", s, flags=re.M | re.S)
assert m
s = s[:m.start()] + para + s[m.end():]
ideal = " / ".join(f"{kt(c, 'k_ideal_profit') / 1e3:.2f}" for c in ("cfg2", "cfg3", "cfg4"))
s, n = re.subn(r"(\| episode-start ideal profit, `k_ideal_profit`: cfg2 / cfg3 / cfg4 \| \*\*)[0-9. /]+( ms\*\*)", lambda mm: mm.group(1) + ideal + mm.group(2), s)
assert n == 1
coll = b["also"]["cfg4"]["collective"]["ms_per_call"]
replace_row("the collective at one rank (cfg4, once per 60 steps)", f"| the collective at one rank (cfg4, once per 60 steps) | own reduction kernels {coll['own_reduction_kernels']:.3f} ms + `ncclAllReduce` of 3080 doubles {coll['allreduce'] * 1e3:.1f} µs |")
cb = b["cpu_baseline"]
replace_row("CPU: C oracle on the box's host", f"| CPU: C oracle on the box's host, 1 core / {cb['cores']} cores (OpenMP over envs) | {sci(cb['single_thread']['value'])} / {sci(cb['value'])} keyword-steps/s |")
open("DESIGN.md", "w").write(s)
print("DESIGN.md section 6 regenerated from", P + "bench.json")
