#!/usr/bin/env python3
"""a binding-budget run of cfg2's law at a scaled click rate, to put under `rocprofv3 --kernel-trace --stats`:
  python3 tools/profile_binding_ctr.py <ctr factor> <budget> [steps]"""
import sys

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd._ffi import P_BCTR  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

f, budget = float(sys.argv[1]), float(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
planes[P_BCTR] *= f
eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(planes)
eng.reset()
eng.sample_actions(0.30, 1.00, budget)
for _ in range(steps):
    eng.step_device()
eng.synchronize()
print(eng.step_kernel_name(), eng.walk_stats().tolist())
eng.close()
