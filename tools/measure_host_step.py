#!/usr/bin/env python3
"""PCIe-inclusive rate of the host boundary (adc_engine_step: bids/budget H2D, observations D2H, pageable numpy
buffers) next to the device-resident rate, on cfg2.  Never the bench `value`; quoted in DESIGN.md."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

N, K, mv, cvr, nvp, drift = synthetic.CONFIGS["cfg2"]
e = StepEngine(N, K, seed=1729, loss_threshold=1e12, auto_reset=True)
e.set_all_params(synthetic.implicit_keyword_planes(N, K, 1729))
e.reset()
bids = np.random.default_rng(0).uniform(0.3, 1.0, (N, K)).astype(np.float32)
budget = np.full(N, 1e9, np.float32)
for _ in range(5):
    e.step(bids, budget, copy=False)
t0 = time.perf_counter()
n = 50
for _ in range(n):
    e.step(bids, budget, copy=False)
host = (time.perf_counter() - t0) / n
e.sample_actions()
for _ in range(5):
    e.step_device()
e.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    e.step_device()
e.synchronize()
dev = (time.perf_counter() - t0) / n
print(json.dumps({"workload": "cfg2 4096x256", "host_step_ms": host * 1e3, "host_U_per_s": N * K / host,
                  "device_step_ms": dev * 1e3, "device_U_per_s": N * K / dev,
                  "pcie_bytes_per_step": N * K * 24 + N * 30}))
