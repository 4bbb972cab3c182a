#!/usr/bin/env python3
"""step time of the reference's default ImplicitKeyword (ADC_MODEL_IMPLICIT_GENERAL: Binomial bidders per auction, top-(w+n)
clearing) - it runs on the reference-order walker, one wavefront per env"""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd._ffi import MODEL_IMPLICIT_GENERAL  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

for N, K in ((1, 100), (256, 100), (4096, 256)):
    planes = synthetic.implicit_keyword_planes(N, K, seed=3)
    for budget in (1e9, 1000.0):
        eng = StepEngine(N, K, MODEL_IMPLICIT_GENERAL, seed=3, max_days=1 << 30, loss_threshold=1e15)
        eng.set_all_params(planes)
        eng.reset()
        eng.sample_actions(0.30, 1.00, budget)
        for _ in range(3):
            eng.step_device()
        eng.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            eng.step_device()
        eng.synchronize()
        print(f"IMPLICIT_GENERAL {N} x {K}, budget {budget:g}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
        eng.close()
