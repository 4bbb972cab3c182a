#!/usr/bin/env python3
"""Runs ON THE GPU BOX: hardware counters of k_step_implicit_fast for one bench config, as a JSON record for
profiles/pmc_traffic.json (what bench.py quotes as `roofline.traffic` / `roofline_valu`, with its source).

  python3 tools/pmc_collect.py <cfg> <outdir>

Three separate rocprofv3 --pmc passes of `python3 bench.py --config <cfg>` (FETCH_SIZE and WRITE_SIZE cannot share a pass;
MI355X_MICROARCH.md, rocprofv3 PMC slots), kernel-trace options only.  HBM bytes = 2 x FETCH_SIZE (gfx950 reports half the
bytes of a coalesced stream; same guide, HBM section) + WRITE_SIZE, in KB of 1024 B, mean over dispatches."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

cfg, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"]}
acc = collections.defaultdict(list)
for name, counters in PASSES.items():
    d = os.path.join(out, f"{cfg}_{name}")
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "pmc", "--", "python3", "bench.py", "--config", cfg,
           "--steps", "12", "--warmup", "2", "--spin-seconds", "0", "--no-cpu-baseline"]
    with open(os.path.join(out, f"{cfg}_{name}.log"), "w") as log:
        # (one env group: every dispatch of the kernel is the full batch - the launch `roofline.kernel_ms` and the algorithmic bytes are per)
        subprocess.check_call(cmd, stdout=log, stderr=subprocess.STDOUT, env=dict(os.environ, ADCRAFT_STREAM_GROUPS="1"))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_step_implicit" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
sys.path.insert(0, ".")
from adcraft_amd import build as hip_build, synthetic  # noqa: E402
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
auctions = N * K * mean_volume * (1.0 - no_vol_prob)
rec = {
    "hbm_bytes_per_launch": 2.0 * m["FETCH_SIZE"] * 1024.0 + m["WRITE_SIZE"] * 1024.0,
    "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"],
    "algorithmic_bytes_per_launch": N * K * (68 if drift else 56) + 26 * N,
    "algorithmic_bytes_incl_metric_mode": N * K * ((68 if drift else 56) + 8) + 26 * N,
    "valu": {"wave_instructions_per_launch": m["SQ_INSTS_VALU"], "active_inst_valu_quad_cycles": m["SQ_ACTIVE_INST_VALU"],
             "wave_quad_cycles": m["SQ_WAVE_CYCLES"], "salu_instructions": m["SQ_INSTS_SALU"], "lds_instructions": m["SQ_INSTS_LDS"],
             "lds_active_quad_cycles": m["SQ_ACTIVE_INST_LDS"], "grbm_gui_active": m["GRBM_GUI_ACTIVE"], "waves": m["SQ_WAVES"],
             "auctions_per_launch_expected": auctions,
             "valu_lane_instructions_per_auction": m["SQ_INSTS_VALU"] * 64.0 / auctions},
    "library_source_hash": hip_build.source_hash(),      # bench.py quotes these counters only for the build they were taken on
    "source_note": "builder-side rocprofv3 --pmc passes of `python3 bench.py --config " + cfg + "` on an MI355X (tools/pmc_collect.py, round 5); "
                   "one env group (ADCRAFT_STREAM_GROUPS=1: every dispatch is the full batch); FETCH_SIZE doubled per the gfx950 correction; bench runs in metric mode (+8 B per keyword-step of accumulator traffic: a 32-bit word read and written)",
}
with open(os.path.join(out, f"pmc_{cfg}.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
