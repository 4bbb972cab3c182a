#!/usr/bin/env python3
"""GPU box: the episode-start ideal profit (k_ideal_profit, 2048 samples per keyword) at cfg2 and cfg3 size, and the curve build.
A/B against another build: ADCRAFT_HIP_LIB=<path>.  Usage: python tools/measure_ideal_profit.py [cfg ...]"""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

for cfg in (sys.argv[1:] or ["cfg2", "cfg3"]):
    N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
    e = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e12, auto_reset=True)
    e.set_all_params(planes)
    e.reset()
    e.ideal_profit(2048)
    e.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ideal = e.ideal_profit(2048)
    t = (time.perf_counter() - t0) / reps
    line = f"{cfg}: {N} x {K}  ideal_profit(2048) {t * 1e3:8.2f} ms (host call incl. the N x K doubles to the host)  sum {ideal.sum():.6f}"
    if cfg == "cfg2":
        e.synchronize()
        t0 = time.perf_counter()
        e.bid_curves_build(2048)
        e.synchronize()
        line += f"   curve build {(time.perf_counter() - t0) * 1e3:8.2f} ms"
    print(line, flush=True)
    e.close()
