#!/usr/bin/env python3
"""host-side cost of BiddingSimulationVectorEnv.step at cfg2 size (what an RL loop with numpy policies sees)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import gymnasium_kw_utils as utils  # noqa: E402
from adcraft_amd.vector_env import BiddingSimulationVectorEnv  # noqa: E402

N, K = 4096, 256
res = {}
for flat, shards, compact in [(False, 1, False), (True, 1, False), (False, 4, False), (True, 4, False), (False, 1, True), (False, 2, True),
                              (False, 3, True), (False, 4, True)]:
    vec = BiddingSimulationVectorEnv(N, keyword_config=utils.experiment_keyword_config(128, 0.8), num_keywords=K, budget=1e6,
                                     param_sampler="device", flat=flat, engine_shards=shards, compact_counts=compact)
    vec.reset(seed=1)
    act = np.concatenate([np.full((N, 1), 1e6, np.float32), np.full((N, K), 0.8, np.float32)], axis=1) if flat else \
        {"keyword_bids": np.full((N, K), 0.8, np.float32), "budget": np.full(N, 1e6, np.float32)}
    for _ in range(3):
        vec.step(act)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        vec.step(act)
    res[("flat" if flat else "dict") + f"_shards{shards}" + ("_u16counts" if compact else "")] = (time.perf_counter() - t0) / n * 1e3
    if compact and shards == 4:
        buf = vec.action_buffers()
        buf["keyword_bids"][...] = 0.8
        buf["budget"][...] = 1e6
        for _ in range(3):
            vec.step(buf)
        t0 = time.perf_counter()
        for _ in range(n):
            vec.step(buf)
        res["dict_shards4_u16counts_actions_in_place"] = (time.perf_counter() - t0) / n * 1e3
    vec.close()
print(json.dumps({"vector_env_step_ms_4096x256": res}))
