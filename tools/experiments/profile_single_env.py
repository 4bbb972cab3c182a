#!/usr/bin/env python3
"""a few hundred device-resident steps of one env x 100 keywords (cfg2's law), to run under `rocprofv3 --kernel-trace --stats`:
  python3 tools/profile_single_env.py [budget]      (default: non-binding; e.g. 30 for a budget that binds every day)"""
import sys

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1e9
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
e = StepEngine(1, 100, seed=3, max_days=1 << 30, loss_threshold=1e15)
e.set_all_params(planes[:, :1, :100])
e.reset()
e.sample_actions(0.3, 1.0, budget)
for _ in range(300):
    e.step_device()
e.synchronize()
print(e.step_kernel_name(), e.walk_stats().tolist())
e.close()
