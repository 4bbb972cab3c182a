#!/usr/bin/env python3
"""GPU box: the contender lists (k_curve_contenders) of this build against another build of the library (ADCRAFT_HIP_LIB of the second
run), same keywords and curves: counts, grid indices and margin intervals must be identical.
Usage: python tools/experiments/contenders_ab.py <other.so>"""
import os
import subprocess
import sys

import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ".")
    from adcraft_amd import synthetic
    from adcraft_amd.engine import StepEngine
    out = sys.argv[2]
    res = {}
    for tag, (N, K, ns, grid, mv) in {"notebook": (1024, 256, 2048, None, 128), "few": (256, 128, 100, None, 128),
                                      "coarse": (256, 128, 2048, np.arange(0.05, 2.0, 0.05), 128), "grid300": (256, 128, 512, np.arange(0.01, 3.01, 0.01), 40)}.items():
        planes = synthetic.implicit_keyword_planes(N, K, seed=7, mean_volume=mv, cvr=0.8)
        rng = np.random.default_rng(3)
        planes[3, ::7] *= rng.uniform(0.2, 3.0, planes[3, ::7].shape).astype(np.float32)      # competitor law spread
        e = StepEngine(N, K, seed=11, drift_enabled=True, max_days=60, loss_threshold=1e12, auto_reset=True)
        e.set_all_params(planes); e.reset()
        if grid is None: e.bid_curves_build(ns)
        else: e.bid_curves_build(ns, grid)
        n, idx, iv = e.bid_curves_contenders()
        res[tag + "_n"], res[tag + "_idx"], res[tag + "_iv"] = n, idx, iv
        e.close()
    np.savez_compressed(out, **res)
else:
    a, b = "/tmp/cont_this.npz", "/tmp/cont_other.npz"
    subprocess.check_call([sys.executable, __file__, "--one", a])
    subprocess.check_call([sys.executable, __file__, "--one", b], env=dict(os.environ, ADCRAFT_HIP_LIB=sys.argv[1], ADCRAFT_ALLOW_STALE_LIB="1"))
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        x, y = A[k], B[k]
        if k.endswith("_n"):
            same = np.array_equal(x, y)
            print(f"{k}: keywords {x.size}, mean contenders {x[x != 65535].mean():.1f} / {y[y != 65535].mean():.1f}, whole-grid {int((x == 65535).sum())} / {int((y == 65535).sum())}, identical {same}, differing keywords {int((x != y).sum())}")
        else:
            n = A[k.rsplit("_", 1)[0] + "_n"].astype(np.int64); n[n == 65535] = 0
            mask = np.arange(x.shape[2])[None, None, :] < n[..., None]
            same = np.array_equal(np.where(mask[..., None] if x.ndim == 4 else mask, x, 0).view(np.uint32 if x.dtype == np.float32 else x.dtype),
                                  np.where(mask[..., None] if y.ndim == 4 else mask, y, 0).view(np.uint32 if y.dtype == np.float32 else y.dtype)) if np.array_equal(A[k.rsplit("_", 1)[0] + "_n"], B[k.rsplit("_", 1)[0] + "_n"]) else False
            print(f"{k}: identical {same}")
        bad += 0 if same else 1
    print("ALL IDENTICAL" if bad == 0 else f"{bad} arrays differ")
    sys.exit(1 if bad else 0)
