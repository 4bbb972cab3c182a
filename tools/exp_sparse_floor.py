#!/usr/bin/env python3
"""Developer tool (GPU box): what bounds the sparse-volume step.  cfg3-sized engine, kernel time by HIP events for
  normal     the cfg3 law
  novolume   every keyword's volume forced to 0 (no auction at all: loads, volume draw, outputs only)
  nometric   the cfg3 law without the metric-mode accumulators
  live_only  no_vol_prob 0 at half the keywords per env (same auctions, no empty keywords)
Usage: python tools/exp_sparse_floor.py [steps]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def run(tag, N, K, planes, metrics=True):
    eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1.0e12, auto_reset=True)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, 1e9)
    eng.metrics_enable(metrics)
    for _ in range(10):
        eng.step_device()
    eng.synchronize()
    eng.profile_enable(True)
    eng.profile_read()
    for _ in range(steps):
        eng.step_device()
    eng.synchronize()
    kernel_ms, launches = eng.profile_read()
    ms = kernel_ms[0] / launches
    b = N * K * 56 + 26 * N
    print(f"{tag:10s} N={N} K={K}  fast kernel {ms:.4f} ms   {b / ms / 1e6:.0f} GB/s algorithmic = {100 * b / ms / 1e6 / 8000:.1f} % of HBM peak", flush=True)
    eng.close()


N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg3"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
run("normal", N, K, planes)
run("nometric", N, K, planes, metrics=False)
z = planes.copy()
z[0] = 0.0
z[1] = 0.0
run("novolume", N, K, z)
run("novol_nom", N, K, z, metrics=False)
half = synthetic.implicit_keyword_planes(N, K // 2, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=0.0)
run("live_only", N, K // 2, half)
