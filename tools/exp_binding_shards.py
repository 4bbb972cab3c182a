#!/usr/bin/env python3
"""GPU box: does running the budget-exact kernels of several env SHARDS side by side (one engine and stream per shard) beat one engine?
Device-resident steps.  Usage: python tools/exp_binding_shards.py N K budget [mean_volume]"""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import ShardedStepEngine, StepEngine  # noqa: E402

N, K, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=128, cvr=0.8, no_vol_prob=0.0)
out = []
for shards in (1, 2, 4):
    eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15) if shards == 1 else \
        ShardedStepEngine(N, K, shards=shards, seed=1729, max_days=1 << 30, loss_threshold=1e15)
    eng.set_all_params(planes)
    eng.reset()
    parts = [eng] if shards == 1 else eng.parts
    for p in parts:
        p.sample_actions(0.30, 1.00, budget)
    for _ in range(3):
        for _ in range(4):
            eng.step_device()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        eng.step_device()
    eng.synchronize()
    out.append(f"{shards}: {(time.perf_counter() - t0) / 40 * 1e3:.4f}")
    eng.close()
print(f"{N} x {K} budget {budget:g}: ms/step by shards  " + "   ".join(out), flush=True)
