#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile_gpu.sh) into a small markdown summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


print(f"# rocprofv3 summary ({out})\n")
# --- kernel stats
stats = list(rows("stats/**/*kernel_stats.csv"))
if stats:
    print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("(anonymous namespace)", "anon").split("(")[0][-60:]
        print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
              f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()
trace = list(rows("stats/**/*kernel_trace.csv"))
if trace:
    r = next((t for t in trace if "implicit_fast" in t["Kernel_Name"]), trace[0])
    keys = [k for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r]
    print("## dispatch shape of k_step_implicit_fast\n")
    print(", ".join(f"{k}={r[k]}" for k in keys), "\n")

# --- counters
for sub, title in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", "SQ counters"), ("pmc_sq2", "SQ counters 2")):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        acc[r["Kernel_Name"].replace("(anonymous namespace)", "anon").split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        continue
    print(f"## {title} (per dispatch, mean over dispatches)\n")
    print("| kernel | counter | mean | n |")
    print("|---|---|---|---|")
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            print(f"| {k} | {c} | {sum(v)/len(v):.6g} | {len(v)} |")
    print()
