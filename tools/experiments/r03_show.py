#!/usr/bin/env python3
"""digest of a full bench.py line: headline, the `also` configs, host step, CPU baseline"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"headline {d['config']['workload'][:40]}: {d['value']:.4g} U/s  {d['ms_per_step']:.4f} ms/step  kernel {r['kernel_ms']:.4f} ms  frac {r['frac']:.4f}  "
      f"event records in timed region {r.get('event_records_in_timed_region')}")
for k, v in d.get("also", {}).items():
    print(f"  {k}: {v['value']:.4g} U/s  {v['ms_per_step']:.4f} ms/step  kernel {v['roofline']['kernel_ms']:.4f} ms  frac {v['roofline']['frac']:.4f}")
if "host_step" in d:
    print("  host_step", {k: d["host_step"][k] for k in ("ms", "ms_u16_counts") if k in d["host_step"]})
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"  cpu_baseline {c['value']:.4g} U/s on {c['cores']} cores; single thread {c['single_thread']['value']:.4g}")
