#!/usr/bin/env python3
"""Developer tool: where the wave-cycles of k_step_implicit_fast go (needs a -DADC_EXP_TIMING build: python adcraft_amd/build.py --timing).
Usage: ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/timing.so python tools/exp_fast_timing.py [config]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import _ffi, synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1.0e12, drift_enabled=drift, auto_reset=True)
eng.set_all_params(planes)
eng.reset()
eng.sample_actions(0.30, 1.00, 1e9)
L = _ffi.lib()
L.adc_debug_read.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
for _ in range(5):
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
v = np.array(list(out), dtype=np.float64)[8:]
waves = v[6]
print(cfg, "kernel ms per step [fast, tail+rows, metric]:", [round(float(x) / steps, 4) for x in kernel_ms])
names = ["phase 1 (loads, volume, law, prefix)", "  of which law setup", "item search", "phase 2 (stage A + B)", "  of which stage B", "phase 3 (outputs)"]
tot = v[0] + v[3] + v[5]
for i, n in enumerate(names):
    print(f"{n:40s} {v[i] / waves:10.0f} cycles per wave   {100 * v[i] / tot:5.1f} %")
print(f"stage-B batches per wave {v[7] / waves:.2f}; cycles per batch {v[4] / max(v[7], 1):.0f}; waves per step {waves / steps:.0f}")
