#!/usr/bin/env python3
"""binding-budget step times for keyword counts beyond one lane per keyword (K = 1024: the cfg4 / cfg5 shapes)"""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

for N, K in ((2048, 1024), (4096, 512)):
    planes = synthetic.implicit_keyword_planes(N, K, seed=11)
    for budget in (1e9, 4000.0, 40.0):
        eng = StepEngine(N, K, seed=11, max_days=1 << 30, loss_threshold=1e15)
        eng.set_all_params(planes)
        eng.reset()
        eng.sample_actions(0.30, 1.00, budget)
        for _ in range(3):      # (between launches the host learns what the device found out)
            for _ in range(3):
                eng.step_device()
            eng.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            eng.step_device()
        eng.synchronize()
        print(f"IMPLICIT {N} x {K}, budget {budget:g}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
        eng.close()
