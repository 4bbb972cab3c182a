#!/usr/bin/env python3
"""step latency of the single-env facade and of the raw engine at small sizes (launch/copy-bound regime)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import adcraft_amd  # noqa: E402
from adcraft_amd import gymnasium_kw_utils as utils  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402
from adcraft_amd import synthetic  # noqa: E402

res = {}
for K in (100,):
    env = adcraft_amd.BiddingSimulation(keyword_config=utils.experiment_keyword_config(128, 0.8), num_keywords=K, budget=1e6)
    env.reset(seed=1)
    act = {"keyword_bids": np.full(K, 0.8, np.float32), "budget": np.float32(1e6)}
    for _ in range(20):
        env.step(act)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        o, r, te, tr, info = env.step(act)
        if te or tr:
            env.reset()
    res[f"facade_step_us_K{K}"] = (time.perf_counter() - t0) / n * 1e6
    env.close()
for N, K in ((1, 100), (64, 100), (1024, 64)):
    e = StepEngine(N, K, seed=1, loss_threshold=1e12, max_days=1 << 30)
    e.set_all_params(synthetic.implicit_keyword_planes(N, K, 1))
    e.reset()
    bids = np.full((N, K), 0.8, np.float32)
    for _ in range(20):
        e.step(bids, 1e9, copy=False)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        e.step(bids, 1e9, copy=False)
    res[f"engine_step_us_{N}x{K}"] = (time.perf_counter() - t0) / n * 1e6
    e.sample_actions()
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        e.step_device()
    e.synchronize()
    res[f"engine_step_device_us_{N}x{K}"] = (time.perf_counter() - t0) / n * 1e6
    e.close()
print(json.dumps(res))
