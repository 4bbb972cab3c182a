import sys, time
import numpy as np
sys.path.insert(0, ".")
from adcraft_amd.engine import StepEngine
from tests import helpers as H
"""EXPLICIT-model (default constructor) step time at several sizes; budget 1000 (the reference default; it binds at
K = 256) and a non-binding budget."""
for N, K, budget in [(1, 64, 1000.0), (256, 64, 1000.0), (4096, 64, 1000.0), (4096, 256, 1000.0), (4096, 256, 1.0e9), (16384, 1024, 1.0e9)]:
    planes = H.explicit_params(N, K, seed=3)
    e = StepEngine(N, K, model=1, seed=5, max_days=1 << 30, loss_threshold=1e12)
    e.set_all_params(planes); e.reset()
    e.sample_actions(0.3, 1.0, budget)
    for _ in range(3): e.step_device()
    e.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): e.step_device()
    e.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"EXPLICIT N={N} K={K} budget={budget:g}: {dt*1e3:.3f} ms/step  {N*K/dt:.3e} U/s")
    e.close()
