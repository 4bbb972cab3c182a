#!/usr/bin/env python3
"""Runs ON THE GPU BOX: SQ counters of one kernel (name substring) under `python3 bench.py <args>`, mean per dispatch.

  python3 tools/pmc_kernel.py <kernel substring> <outdir> [bench.py args ...]

Two separate rocprofv3 --pmc passes (kernel-trace options only; MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

kern, out, bench_args = sys.argv[1], sys.argv[2], sys.argv[3:]
os.makedirs(out, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
PASSES = {"sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"],
          "wait": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SMEM", "SQ_BUSY_CYCLES"]}
acc = collections.defaultdict(list)
for name, counters in PASSES.items():
    d = os.path.join(out, name)
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "pmc", "--", "python3", "bench.py", *bench_args]
    with open(os.path.join(out, f"{name}.log"), "w") as log:
        subprocess.check_call(cmd, stdout=log, stderr=subprocess.STDOUT)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    subprocess.call(["rm", "-rf", d])
m = {k: sum(v) / len(v) for k, v in acc.items()}
m["dispatches"] = max((len(v) for v in acc.values()), default=0)
with open(os.path.join(out, "pmc_kernel.json"), "w") as f:
    json.dump({"kernel": kern, "bench_args": bench_args, "mean_per_dispatch": m}, f, indent=1)
print(json.dumps(m, indent=1))
