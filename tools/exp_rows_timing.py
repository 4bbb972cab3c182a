#!/usr/bin/env python3
"""Developer tool: phase times inside k_step_exact_rows (needs a -DADC_EXP_TIMING build passed via ADCRAFT_HIP_LIB).
Usage: ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/timing.so python tools/exp_rows_timing.py [budget] [config]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import _ffi, synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1.0e12, drift_enabled=drift, auto_reset=True)
eng.set_all_params(planes)
eng.reset()
eng.sample_actions(0.30, 1.00, budget)
L = _ffi.lib()
L.adc_debug_read.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
for _ in range(4):          # (the host learns what the device found out - budgets bind, envs to park at once - between launches)
    for _ in range(3):
        eng.step_device()
    eng.synchronize()
L.adc_debug_read(out, 1)
eng.profile_enable(True)
eng.profile_read()
steps = 20
for _ in range(steps):
    eng.step_device()
eng.synchronize()
kernel_ms, launches = eng.profile_read()
L.adc_debug_read(out, 0)
v = np.array(list(out), dtype=np.float64)
print("kernel ms per step [fast, tail, rows]:", [round(float(x) / steps, 4) for x in kernel_ms])
print(f"non-binding rows/env-day {v[7] / max(v[4], 1):.2f}, their resolve (= chain) {v[2] * 10 / max(v[7], 1) / 1e3:.2f} us/row; "
      f"binding rows resolve {(v[1] - v[2]) * 10 / max(v[5] - v[7], 1) / 1e3:.2f} us/row")
print(f"env-days/step {v[4] / steps:.0f}  rows/env-day {v[5] / max(v[4], 1):.2f}  walker calls/row {v[6] / max(v[5], 1):.2f}")
for i, n in enumerate(["passA", "resolve (incl. chain)", "  chain + its barrier", "passB"]):
    print(f"{n:24s} {v[i] * 10 / max(v[5], 1) / 1e3:8.2f} us/row   {v[i] * 10e-9 * 1e3 / steps:8.2f} block-ms/step")
eng.walk_stats()      # (timing build: prints the phase sums of k_step_click_walk and k_step_rest_of_day to stderr)
print("env-days parked at once (adc_debug_direct_days):", eng.direct_days())
