#!/usr/bin/env python3
"""A few dozen device-resident IMPLICIT steps of N x K at a budget, to put under `rocprofv3 --kernel-trace --stats`.
usage: profile_shape.py N K budget [steps]"""
import sys

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

N, K, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
planes = synthetic.implicit_keyword_planes(N, K, seed=11)
eng = StepEngine(N, K, seed=11, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(planes)
eng.reset()
eng.sample_actions(0.30, 1.00, budget)
for _ in range(3):
    for _ in range(3):
        eng.step_device()
    eng.synchronize()
for _ in range(steps):
    eng.step_device()
eng.synchronize()
print(eng.step_kernel_name())
eng.close()
