#!/usr/bin/env python3
"""one-line summary of a bench.py JSON line read from stdin (tag = argv[1])"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "U/s=%.4g" % d["value"], "ms/step=%.3f" % d["ms_per_step"],
      "fast_kernel_ms=%.3f" % r["kernel_ms"], "GB/s=%.0f frac=%.4f" % (r["achieved"], r["frac"]))
