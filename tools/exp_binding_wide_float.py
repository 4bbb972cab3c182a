#!/usr/bin/env python3
"""binding-budget step times of the float-money models beyond 256 keywords (the 512- / 1024-lane day kernel):
EXPLICIT and the default ImplicitKeyword at 4096 x 512 and 2048 x 1024.  A/B against another build: ADCRAFT_HIP_LIB=<path>."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd.engine import MODEL_EXPLICIT, StepEngine  # noqa: E402
from tests import helpers as H  # noqa: E402


def ms(eng, n=12):
    for _ in range(4):
        eng.step_device()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step_device()
    eng.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for N, K in ((4096, 512), (2048, 1024)):
    xp = H.explicit_params(N, K, seed=5)
    for budget in (1e9, 4.0 * K):
        eng = StepEngine(N, K, MODEL_EXPLICIT, seed=5, max_days=1 << 30, loss_threshold=1e15)
        eng.set_all_params(xp)
        eng.reset()
        eng.sample_actions(0.30, 1.00, budget)
        print(f"EXPLICIT {N} x {K}, budget {budget:g}: {ms(eng):.3f} ms/step", flush=True)
        eng.close()
    rng = np.random.default_rng(8)
    gp = np.stack([rng.integers(20, 120, (N, K)), rng.random((N, K)) * 6, rng.uniform(0.0, 0.3, (N, K)), rng.uniform(0.05, 0.15, (N, K)),
                   rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.3, 1.5, (N, K)),
                   rng.uniform(0.02, 0.3, (N, K))]).astype(np.float32)
    for budget in (1e9, 0.25 * K):
        eng = StepEngine(N, K, 2, seed=5, max_days=1 << 30, loss_threshold=1e15)
        eng.set_general_model(30, 0.6, 1)
        eng.set_all_params(gp)
        eng.reset()
        eng.sample_actions(0.05, 0.50, budget)
        print(f"IMPLICIT_GENERAL {N} x {K}, budget {budget:g}: {ms(eng):.3f} ms/step", flush=True)
        eng.close()
