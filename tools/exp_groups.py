#!/usr/bin/env python3
"""GPU box: step time of ONE engine by the number of env groups it runs an IMPLICIT step as (ADCRAFT_STREAM_GROUPS; 0 = the engine's
choice), device-resident steps.  Usage: python tools/exp_groups.py N K budget [mean_volume cvr no_vol_prob]"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ".")
    from adcraft_amd import synthetic
    from adcraft_amd.engine import StepEngine
    N, K, budget = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    mv, cvr, nv = (float(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])) if len(sys.argv) > 7 else (128.0, 0.8, 0.0)
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mv, cvr=cvr, no_vol_prob=nv)
    eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e15, auto_reset=True)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, budget)
    eng.metrics_enable(True)
    for _ in range(3):
        for _ in range(6):
            eng.step_device()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        eng.step_device()
    eng.synchronize()
    print(f"{(time.perf_counter() - t0) / 60 * 1e3:.4f}")
    eng.close()
else:
    out = []
    for g in ("1", "2", "4", "0"):
        env = dict(os.environ, ADCRAFT_STREAM_GROUPS=g)
        r = subprocess.run([sys.executable, __file__, "--one", *sys.argv[1:]], env=env, capture_output=True, text=True)
        out.append(f"{g}: {r.stdout.strip() or r.stderr.strip()[-200:]}")
    print(" ".join(sys.argv[1:4]) + " ms/step by groups  " + "   ".join(out), flush=True)
