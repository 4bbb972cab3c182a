#!/usr/bin/env python3
"""EXPLICIT 4096 x 256 at a binding budget: a few dozen device-resident steps (run under rocprofv3 --kernel-trace --stats)"""
import sys

sys.path.insert(0, ".")
from adcraft_amd.engine import MODEL_EXPLICIT, StepEngine  # noqa: E402
from tests import helpers as H  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
N, K = 4096, 256
eng = StepEngine(N, K, MODEL_EXPLICIT, seed=5, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(H.explicit_params(N, K, seed=5))
eng.reset()
eng.sample_actions(0.30, 1.00, budget)
for _ in range(40):
    eng.step_device()
eng.synchronize()
out = eng.fetch()
print("clicks per env-day", out["buyside_clicks"].sum(axis=1).mean(), "spend per env-day", out["cost"].sum(axis=1).mean())
