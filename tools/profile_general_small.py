#!/usr/bin/env python3
"""GPU box: a few hundred steps of ONE env x 100 keywords of the default ImplicitKeyword (ADC_MODEL_IMPLICIT_GENERAL), to put under
rocprofv3 --kernel-trace --stats.  Usage: python3 tools/profile_general_small.py [budget] [lanes-per-keyword path 0/1]"""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd._ffi import MODEL_IMPLICIT_GENERAL  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1e9
N, K = 1, 100
eng = StepEngine(N, K, MODEL_IMPLICIT_GENERAL, seed=3, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(synthetic.implicit_keyword_planes(N, K, seed=3))
eng.reset()
eng.sample_actions(0.30, 1.00, budget)
for _ in range(20):
    eng.step_device()
eng.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    eng.step_device()
eng.synchronize()
print(f"IMPLICIT_GENERAL {N} x {K}, budget {budget:g}: {(time.perf_counter() - t0) / n * 1e6:.1f} us/step ({eng.step_kernel_name()})", flush=True)
eng.close()
