#!/usr/bin/env python3
"""binding-budget step times: cfg2 at budgets 1000 / 10 / 1, one env x 100 keywords, EXPLICIT 4096 x 256"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import MODEL_EXPLICIT, MODEL_IMPLICIT, StepEngine  # noqa: E402


def device_ms(eng, steps=60):
    for _ in range(3):          # (between launches the host learns what the device found out: budgets bind, envs list their clicks, ...)
        for _ in range(4):
            eng.step_device()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step_device()
    eng.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
for budget in (1e9, 1000.0, 10.0, 1.0):
    eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, budget)
    eng.walk_stats(reset=True)
    print(f"cfg2 IMPLICIT 4096 x 256, budget {budget:g}: {device_ms(eng):.3f} ms/step   click walk [walked, overflowed, stopped, other] = "
          f"{eng.walk_stats().tolist()}", flush=True)
    eng.close()
eng = StepEngine(1, 100, seed=3, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(planes[:, :1, :100])
eng.reset()
for budget in (1e9, 30.0):
    eng.sample_actions(0.30, 1.00, budget)
    print(f"1 env x 100 keywords, budget {budget:g}: {device_ms(eng, 200) * 1e3:.1f} us/step (device-resident)", flush=True)
    bids, bud = np.full((1, 100), 0.8, np.float32), np.full(1, budget, np.float32)
    for _ in range(10):
        eng.step(bids, bud, copy=False)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.step(bids, bud, copy=False)
    print(f"1 env x 100 keywords, budget {budget:g}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us/step (host in / host out)   click walk "
          f"{eng.walk_stats(reset=True).tolist()}", flush=True)
eng.close()
from tests import helpers as H  # noqa: E402
xp = H.explicit_params(N, K, seed=5)
if True:
    for budget in (1e9, 1000.0):
        eng = StepEngine(N, K, MODEL_EXPLICIT, seed=5, max_days=1 << 30, loss_threshold=1e15)
        eng.set_all_params(xp)
        eng.reset()
        eng.sample_actions(0.30, 1.00, budget)
        print(f"EXPLICIT 4096 x 256, budget {budget:g}: {device_ms(eng, 30):.3f} ms/step", flush=True)
        eng.close()
