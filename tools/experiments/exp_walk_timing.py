#!/usr/bin/env python3
"""where k_step_click_walk spends its time (timing build: python adcraft_amd/build.py --timing; run with
ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/timing.so): the phase sums are printed to stderr by adc_debug_walk_stats"""
import sys

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(planes)
eng.reset()
eng.sample_actions(0.30, 1.00, budget)
for _ in range(6):
    eng.step_device()
eng.synchronize()
eng.walk_stats(reset=True)
steps = 20
for _ in range(steps):
    eng.step_device()
eng.synchronize()
print(f"budget {budget:g}: stats {eng.walk_stats().tolist()} over {steps} steps of {N} envs "
      "(stderr: ticks of 10 ns summed over env-steps: setup | sort | walk | commit | gather | second part | groups | candidates)")
