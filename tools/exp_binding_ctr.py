#!/usr/bin/env python3
"""binding budgets at lower click rates: cfg2's keyword law with the buy-side CTR scaled down (the reference's quantile-based
generators emit CTRs of a few percent; cfg2's experiment law has 0.5), budget scaled alike so that it binds in the same part of
the day; the click-list walk (k_step_click_walk) against the row kernel alone (ADCRAFT_CLICK_WALK=0)"""
import os
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd._ffi import P_BCTR  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402


def device_ms(eng, steps=60):
    for _ in range(8):
        eng.step_device()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step_device()
    eng.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg2"]
base = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
for f in (1.0, 0.3, 0.1, 0.03):
    planes = base.copy()
    planes[P_BCTR] *= f
    for budget in (1000.0 * f, 10.0 * f):
        res = {}
        for walk in ("1", "0"):
            os.environ["ADCRAFT_CLICK_WALK"] = walk
            if len(sys.argv) > 1:
                os.environ["ADCRAFT_CLICK_WALK_MAX"] = sys.argv[1]
            eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15)
            eng.set_all_params(planes)
            eng.reset()
            eng.sample_actions(0.30, 1.00, budget)
            eng.walk_stats(reset=True)
            res[walk] = (device_ms(eng), eng.walk_stats().tolist())
            eng.close()
        print(f"cfg2 law, CTR x {f:g}, budget {budget:g}: click walk {res['1'][0]:.3f} ms/step {res['1'][1]}   row kernel alone {res['0'][0]:.3f} ms/step", flush=True)
