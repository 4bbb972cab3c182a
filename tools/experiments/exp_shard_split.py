#!/usr/bin/env python3
"""host step of 4096 x 256 against the way the envs are split over engines (equal parts, growing parts)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import ShardedStepEngine  # noqa: E402

N, K = 4096, 256
planes = synthetic.implicit_keyword_planes(N, K, seed=1)
bids = np.full((N, K), 0.8, np.float32)
budget = np.full(N, 1e6, np.float32)
for compact in (False, True):
    for split in (4, (1, 3, 4, 8), (1, 1, 2, 4, 8), (1, 2, 3, 4, 6), (1, 1, 2, 4, 8, 16), (1, 2, 4, 8, 8, 9), (2, 3, 3, 4, 4)):
        s = ShardedStepEngine(N, K, shards=split, seed=1, compact_counts=compact)
        s.set_all_params(planes)
        s.reset()
        for _ in range(5):
            s.step(bids, budget, copy=False)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                s.step(bids, budget, copy=False)
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        print(f"u16counts={compact} split={split}: {best:.3f} ms", flush=True)
        s.close()
