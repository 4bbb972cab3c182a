#!/usr/bin/env python3
"""GPU box: the per-step ideal at cfg2 size (4096 x 256, 299-point grid): contender lists against the whole grid, and
the one-off cost of building the lists.  Usage: python tools/measure_ideal_step.py"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from adcraft_amd import synthetic
    from adcraft_amd.engine import StepEngine
    N, K = 4096, 256
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=128, cvr=0.8)
    e = StepEngine(N, K, seed=1729, drift_enabled=True, max_days=60, loss_threshold=1e12, auto_reset=True)
    e.set_all_params(planes)
    e.reset()
    e.synchronize()
    t0 = time.perf_counter()
    e.bid_curves_build(2048)
    e.synchronize()
    t_build = time.perf_counter() - t0
    e.sample_actions(0.3, 1.0, 1e9)
    for _ in range(5):
        e.ideal_step(fetch=False)
        e.step_device()
    e.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        e.ideal_step(fetch=False)
    e.synchronize()
    t_ideal = (time.perf_counter() - t0) / n
    nc = None
    print(f"{sys.argv[1]:28s} curve build {t_build * 1e3:8.2f} ms   ideal_step {t_ideal * 1e3:7.4f} ms")
    e.close()
else:
    for name, env in (("whole grid (checker)", {"ADCRAFT_IDEAL_FULL_SCAN": "1"}), ("contender lists", {"ADCRAFT_IDEAL_FULL_SCAN": "0"})):
        subprocess.check_call([sys.executable, __file__, name], env=dict(os.environ, **env))
