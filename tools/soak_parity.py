#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: random shapes, keyword laws, budgets, drift, autoreset; every step compared
bit for bit.  Usage: python tools/soak_parity.py [seconds] [seed] [explicit | general | lists]
(lists: IMPLICIT engines of at most 256 keywords stepped 6-14 times with budgets that keep binding, so that the click lists of
k_step_click_walk - which an env starts on its third binding day - and k_step_rest_of_day are what is compared most of the time)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd.engine import StepEngine  # noqa: E402
from oracle import capi as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
explicit = len(sys.argv) > 3 and sys.argv[3] == "explicit"      # the default-constructor (ExplicitKeyword) model
general = len(sys.argv) > 3 and sys.argv[3] == "general"        # the default ImplicitKeyword (bidder pools, top-(w+n) clearing)
lists = len(sys.argv) > 3 and sys.argv[3] in ("lists", "small")
small = len(sys.argv) > 3 and sys.argv[3] == "small"      # budgets that run out within the first cells of the day, day after day (hint 6: parked at once)
t0 = time.time()
cases = steps = reruns = at_once = 0
last_report = t0
while time.time() - t0 < budget_s:
    if time.time() - last_report > 60:          # (a silent GPU job is taken for a hung one)
        print(f"... {cases} engines, {steps} steps so far", flush=True)
        last_report = time.time()
    N = int(rng.integers(1, 9))
    K = int(rng.choice([1, 2, 7, 63, 64, 65, 100, 255, 256, 257, 300, 511, 700, 1024, 1100]))
    mv = float(rng.choice([0, 1, 5, 16, 40, 128, 600]))
    if lists:
        K = int(rng.choice([7, 64, 100, 200, 256]))
        mv = float(rng.choice([16, 40, 128, 300]))
    planes = H.implicit_params(N, K, seed=int(rng.integers(1 << 30)), mean_volume=mv, cvr=float(rng.uniform(0, 1)),
                               no_vol_prob=float(rng.choice([0.0, 0.3, 0.9])))
    if rng.random() < 0.3:
        planes[4] = rng.choice([0.0, 1.0, 0.5], size=planes[4].shape).astype(np.float32)       # extreme click rates
    if rng.random() < 0.2:
        planes[3] *= np.float32(rng.choice([0.0, 5.0]))                                           # degenerate / wide competitor
    if explicit:
        K = min(K, 1100)                          # (the 512- and 1024-lane day kernels, and the walkers beyond)
        planes = H.explicit_params(N, K, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.3:
            planes[0] *= np.float32(rng.choice([0.0, 4.0]))          # no volume / ~100 auctions
    pool = None
    if general:                                   # tens of competitor bids per auction: keep the oracle's share of the time small
        K = min(K, 700)
        planes = H.implicit_params(N, K, seed=int(rng.integers(1 << 30)), mean_volume=float(rng.choice([0, 3, 20, 60])), cvr=float(rng.uniform(0, 1)),
                                   no_vol_prob=float(rng.choice([0.0, 0.3])))
        planes[2] = rng.uniform(0.0, 0.3, planes[2].shape).astype(np.float32)         # competitors' Laplace location ...
        planes[3] = rng.uniform(0.02, 0.2, planes[3].shape).astype(np.float32)        # ... and scale (raw bids, not 1 / scale)
        pool = (int(rng.choice([1, 3, 9, 30, 70])), float(rng.choice([0.2, 0.6, 1.0])), int(rng.integers(1, 3)))
    drift = bool(rng.random() < 0.4)
    auto = bool(rng.random() < 0.5)
    max_days = int(rng.integers(1, 5))
    loss = float(rng.choice([1e9, 20.0]))
    e = StepEngine(N, K, model=1 if explicit else 2 if general else 0, seed=int(rng.integers(1 << 30)), drift_enabled=drift, drift=(0.1, 0.2, 0.3), max_days=max_days,
                   loss_threshold=loss, auto_reset=auto)
    if general:
        e.set_general_model(*pool)
    e.set_all_params(planes)
    e.reset(seeds=rng.integers(0, 1 << 40, N).astype(np.uint64))
    extra = dict(max_bidders=pool[0], participation_rate=pool[1], num_winners=pool[2]) if general else {}
    o = H.mirror_oracle(e, planes, drift_on=drift, drift=(0.1, 0.2, 0.3), max_days=max_days, loss_threshold=loss, auto_reset=auto, **extra)
    day_cost = None          # (lists) what an unconstrained day of this engine costs, per env: budgets are set relative to it
    for s in range(int(rng.integers(6, 15)) if lists else int(rng.integers(1, 6))):
        bids = o.sample_bids(float(rng.uniform(0.01, 0.6)), float(rng.uniform(0.6, 2.0)))
        budget = rng.choice([1e9, 500.0, 50.0, 5.0, 0.3, 0.0], size=N).astype(np.float32)
        if lists and day_cost is not None:
            frac = rng.choice([0.02, 0.2, 0.5, 0.8, 0.97, 3.0], size=N, p=[0.15, 0.25, 0.25, 0.2, 0.1, 0.05])
            if small:
                frac = rng.choice([0.0005, 0.002, 0.006, 0.02, 0.3, 3.0], size=N, p=[0.25, 0.3, 0.25, 0.1, 0.05, 0.05])
            budget = np.maximum(day_cost * frac, 0.01).astype(np.float32)
        elif lists:
            budget = np.full(N, 1e9, dtype=np.float32)
        got, ref = e.step(bids, budget), o.step(bids, budget)
        try:
            H.assert_step_equal(got, ref, implicit=not (explicit or general))
        except AssertionError:
            print("MISMATCH", dict(N=N, K=K, mv=mv, drift=drift, auto=auto, step=s, budget=budget.tolist()))
            raise
        reruns += int((ref["cost_cents"].sum(axis=1) >= np.rint(budget.astype(np.float64) * 100)).sum())
        if lists and day_cost is None:
            day_cost = np.maximum(ref["cost_cents"].sum(axis=1) / 100.0, 1.0)
        steps += 1
    if drift:
        o.materialize_drift()
        assert np.array_equal(e.get_all_params(), o.params)
    if lists:
        walk = e.walk_stats().tolist()          # (the device's counters: cumulative over the process)
        at_once += e.direct_days()          # (a counter per engine)
    e.close()
    cases += 1
if lists:
    print(f"env-days parked at once by k_tail_or_flag: {at_once}")
    print(f"k_step_click_walk: {walk[0]} env-days walked ({walk[2]} of them with a campaign stop), {walk[1]} lists overflowed, {walk[3]} other hand-overs")
print(f"soak ok: {cases} engines, {steps} steps, {reruns} budget-bound env-steps, {time.time() - t0:.0f} s")
