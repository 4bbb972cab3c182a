#!/usr/bin/env python3
"""Single env / few envs with a binding budget (the row kernel alone): device step time."""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd.engine import StepEngine  # noqa: E402
from tests import helpers as H  # noqa: E402

for N, K, mv in [(1, 100, 64), (1, 100, 128), (16, 100, 64), (64, 256, 128)]:
    planes = H.implicit_params(N, K, seed=3, mean_volume=mv)
    e = StepEngine(N, K, seed=5, max_days=1 << 20, loss_threshold=1e12)
    e.set_all_params(planes)
    e.reset()
    for budget in (1e9, 20.0):
        e.sample_actions(0.3, 1.0, budget)
        for _ in range(10):
            e.step_device()
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            e.step_device()
        e.synchronize()
        print(f"N={N} K={K} mean_volume={mv} budget={budget:g}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per device step")
    e.close()
