#!/usr/bin/env python3
"""Throughput of the device-resident baseline loop (agent update + act, per-step ideal profit, env step) on a BASELINE
config; prints one JSON line.  Usage: python tools/measure_closed_loop.py [cfg2] [zero_margin|oracle] [steps] [drift 0|1]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
policy = sys.argv[2] if len(sys.argv) > 2 else "zero_margin"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
drift = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
N, K, mean_volume, cvr, no_vol_prob, _ = synthetic.CONFIGS[cfg]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
e = StepEngine(N, K, seed=1729, max_days=steps, loss_threshold=1e12, drift_enabled=drift)
e.set_all_params(planes)
e.reset()
t0 = time.perf_counter()
e.bid_curves_build(2048)
e.synchronize()
t_curves = time.perf_counter() - t0
e.metrics_enable(True)
e.metrics_reset()
e.agent_init(1.0, None)


def one_step():
    if policy == "zero_margin":
        e.agent_step(100000.0)
        e.ideal_step(fetch=False)
    else:
        e.ideal_step(fetch=False)
        e.policy_oracle(100000.0)
    e.step_device()


for _ in range(3):
    one_step()
e.synchronize()
e.metrics_reset()
t0 = time.perf_counter()
for _ in range(steps):
    one_step()
e.synchronize()
dt = time.perf_counter() - t0
profit, ideal, ideal_pos = e.metrics_read_nk()
ncp = profit.sum(axis=1) / np.maximum(ideal.sum(axis=1), 1e-9)
akncp = np.median(profit / ideal_pos, axis=1)
print(json.dumps(dict(config=cfg, policy=policy, drift=drift, num_envs=N, num_keywords=K, steps=steps,
                      ms_per_loop_step=1e3 * dt / steps, keyword_steps_per_s=N * K * steps / dt,
                      bid_curves_build_s=t_curves, bid_curves_bytes=int(N) * K * 299 * 8,
                      median_NCP=float(np.median(ncp)), median_AKNCP=float(np.median(akncp)))))
