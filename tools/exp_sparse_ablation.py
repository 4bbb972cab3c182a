#!/usr/bin/env python3
"""Developer tool (GPU box): the keyword-parallel kernel's time on one config, for A/B runs of library variants
(ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/<name>.so; ablation builds: adcraft_amd/build.py --variant).
Usage: python tools/exp_sparse_ablation.py [cfg] [steps] [metrics 0/1]"""
import sys

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
metrics = (int(sys.argv[3]) != 0) if len(sys.argv) > 3 else True
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1.0e12, auto_reset=True, drift_enabled=drift)
eng.set_all_params(planes)
eng.reset()
eng.sample_actions(0.30, 1.00, 1e9)
eng.metrics_enable(metrics)
for _ in range(40):
    eng.step_device()
eng.synchronize()
eng.profile_enable(True)
eng.profile_read()
for _ in range(steps):
    eng.step_device()
eng.synchronize()
kernel_ms, launches = eng.profile_read()
import os
print(f"{os.environ.get('ADCRAFT_HIP_LIB', 'product')[-24:]:24s} {cfg} metrics={int(metrics)} {eng.step_kernel_name():28s} {kernel_ms[0] / launches:.4f} ms", flush=True)
eng.close()
