#!/usr/bin/env python3
"""GPU box: does running the budget-exact kernels of several env SHARDS side by side (one engine and stream per shard) beat one engine?
cfg2 at a binding budget, device-resident steps.  Usage: python tools/exp_binding_shards.py [budget]"""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import ShardedStepEngine, StepEngine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
N, K, mv, cvr, nv, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mv, cvr=cvr, no_vol_prob=nv)
for shards in (1, 2, 4, 8):
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
    for _ in range(60):
        eng.step_device()
    eng.synchronize()
    print(f"cfg2 budget {budget:g}, {shards} shard(s): {(time.perf_counter() - t0) / 60 * 1e3:.4f} ms/step", flush=True)
    eng.close()
