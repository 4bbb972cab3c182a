#!/usr/bin/env python3
"""debugging aid: the wave-per-tile sparse kernel against the oracle on the drift case, printing where they differ"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["ADCRAFT_FAST_VARIANT"] = "2"; os.environ["ADCRAFT_FAST_TILE_KW"] = "256"
os.environ["ADCRAFT_SPARSE_WAVE"] = sys.argv[1] if len(sys.argv) > 1 else "2"
os.environ["ADCRAFT_SPARSE_TILES_PER_WAVE"] = "3"
from adcraft_amd.engine import StepEngine
from tests import helpers as H
N, K = 5, 520
planes = H.implicit_params(N, K, seed=38, mean_volume=16, cvr=0.1, no_vol_prob=0.5)
e = StepEngine(N, K, seed=7, drift_enabled=True, max_days=3, loss_threshold=30.0, auto_reset=True)
e.set_all_params(planes); e.reset()
o = H.mirror_oracle(e, planes, drift_on=True, max_days=3, loss_threshold=30.0, auto_reset=True)
for s in range(7):
    bids = o.sample_bids(0.3, 1.0)
    got, ref = e.step(bids, 1e9), o.step(bids, 1e9)
    bad = np.argwhere(got["impressions"] != ref["impressions"])
    badc = np.argwhere(got["buyside_clicks"] != ref["clicks"])
    print("step", s, e.step_kernel_name(), "imp mismatches", len(bad), "click mismatches", len(badc), "volumes", ref["volumes"].sum())
    for env, k in bad[:12]:
        print("   env", env, "kw", k, "got", got["impressions"][env, k], "ref", ref["impressions"][env, k], "V", ref["volumes"][env, k], "tile", k // 128, "lane", (k % 128) // 2)
    o.materialize_drift() if False else None
    gp = e.get_all_params() if False else None
