#!/usr/bin/env python3
"""digest of a bench.py JSON line (file argv[1], or stdin): the headline and every config under "also" """
import json
import sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(src.strip().splitlines()[-1])


def row(name, x):
    r = x["roofline"]
    v = x.get("roofline_valu") or {}
    print(f"{name:6s} U/s={x['value']:.4g} ms/step={x['ms_per_step']:.4f} before_spin={x.get('ms_per_step_before_clock_spin', float('nan')):.4f} "
          f"{r['kernel']} {r['kernel_ms'] * 1e3:.1f} us  HBM {r['achieved']:.0f} GB/s frac={r['frac']:.4f}  traffic={r.get('traffic')} "
          f"valu_frac_4cycle={v.get('frac_4cycle')}")


row("main", d)
for k, x in (d.get("also") or {}).items():
    row(k, x)
if "host_step" in d:
    print("host_step", d["host_step"].get("ms"), d["host_step"].get("ms_u16_counts"))
if "cpu_baseline" in d:
    print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "cores")
