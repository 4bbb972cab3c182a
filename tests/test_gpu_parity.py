"""-m gpu: the HIP engine (through the C ABI) against the CPU oracle and the reference's golden vectors.

Bit-exact for every integer (impressions, clicks, conversions, days, flags) and for money
(integer cents on both sides; the float32 dollars the kernels store are compared against the
identically rounded oracle cents; f64 reward / cumulative profit compared bitwise).
"""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import adcraft_amd.engine as eng
    from adcraft_amd import _ffi
    assert _ffi.device_count() >= 1, "no HIP device visible: the engine has no CPU path"
    return eng


def _run_vs_oracle(amd, N, K, planes, steps, budget, seed=7, model=0, drift=False, max_days=60, loss_threshold=1e4,
                   auto_reset=False, bid_lo=0.3, bid_hi=1.0, check_params=False):
    e = amd.StepEngine(N, K, model=model, seed=seed, drift_enabled=drift, max_days=max_days,
                       loss_threshold=loss_threshold, auto_reset=auto_reset)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=drift, max_days=max_days, loss_threshold=loss_threshold,
                        auto_reset=auto_reset)
    n_rerun = 0
    for s in range(steps):
        bids = o.sample_bids(bid_lo, bid_hi)
        got = e.step(bids, budget)
        ref = o.step(bids, budget)
        H.assert_step_equal(got, ref, implicit=(model == 0))
        n_rerun += int((ref["cost_cents"].sum(axis=1) >= np.rint(np.float64(np.float32(budget)) * 100)).sum())
    if check_params:
        o.materialize_drift()
        assert np.array_equal(e.get_all_params(), o.params)
    e.close()
    return n_rerun


# ------------------------------------------------------------------ PHILOX mode, fast pass
@pytest.mark.parametrize("N,K", [(8, 64), (3, 300), (2, 256), (5, 1), (1, 1024), (4, 257)])
def test_fast_pass_matches_oracle(amd, N, K):
    planes = H.implicit_params(N, K, seed=N * 1000 + K)
    _run_vs_oracle(amd, N, K, planes, steps=3, budget=1.0e9)


def test_sparse_volume_keywords(amd):
    """cfg3 law: half the keywords have (0, U*0.5) volume (gymnasium_kw_utils.py:295-300)"""
    planes = H.implicit_params(6, 320, seed=3, mean_volume=16, cvr=0.1, no_vol_prob=0.5)
    _run_vs_oracle(amd, 6, 320, planes, steps=4, budget=1.0e9)


@pytest.mark.parametrize("case", ["cfg3_law", "ragged_tiles", "beyond_the_item_list", "no_volume", "extreme_rates", "degenerate_laws",
                                  "drift_autoreset", "binding_budget"])
def test_sparse_kernel_forced(amd, monkeypatch, case):
    """k_step_implicit_sparse (bracket classification, item list) on shapes the host's hint would not send to it: forced with
    the engine's scheduling overrides - results may never depend on them"""
    monkeypatch.setenv("ADCRAFT_FAST_VARIANT", "2")
    monkeypatch.setenv("ADCRAFT_FAST_TILE_KW", "256")
    monkeypatch.setenv("ADCRAFT_FAST_TILES_PER_WG", "3")
    if case == "cfg3_law":
        planes = H.implicit_params(7, 1024, seed=31, mean_volume=16, cvr=0.1, no_vol_prob=0.5)
        _run_vs_oracle(amd, 7, 1024, planes, steps=3, budget=1.0e9)
    elif case == "ragged_tiles":
        for N, K in ((5, 257), (3, 300), (4, 1), (2, 255)):
            planes = H.implicit_params(N, K, seed=32 + K, mean_volume=12, cvr=0.5, no_vol_prob=0.3)
            _run_vs_oracle(amd, N, K, planes, steps=2, budget=1.0e9)
    elif case == "beyond_the_item_list":      # > 1024 work items per tile: the listed part, then the search
        planes = H.implicit_params(3, 300, seed=33, mean_volume=128)
        _run_vs_oracle(amd, 3, 300, planes, steps=2, budget=1.0e9)
        planes = H.implicit_params(2, 40, seed=34, mean_volume=3000)
        _run_vs_oracle(amd, 2, 40, planes, steps=2, budget=1.0e9)
    elif case == "no_volume":
        planes = H.implicit_params(3, 300, seed=35)
        planes[0] = 0.0
        planes[1] = 0.0
        _run_vs_oracle(amd, 3, 300, planes, steps=2, budget=1.0e9)
    elif case == "extreme_rates":
        planes = H.implicit_params(4, 260, seed=36, mean_volume=20)
        planes[4, 0] = 1.0
        planes[4, 1] = 0.0
        planes[4, 2] = np.float32(1e-9)
        planes[4, 3] = np.float32(1.0 - 1e-7)
        planes[5, 0] = 0.0
        planes[5, 1] = 1.0
        _run_vs_oracle(amd, 4, 260, planes, steps=3, budget=1.0e9, bid_lo=0.01, bid_hi=2.0)
    elif case == "degenerate_laws":           # brackets that cannot be trusted: everything takes the long way
        planes = H.implicit_params(4, 260, seed=37, mean_volume=20)
        planes[3, 0] = 0.0
        planes[3, 1] *= 50.0
        planes[3, 2] = np.float32(1e-6)
        planes[2, 3] = -0.4
        planes[2, 0, :9] = np.nan
        planes[3, 0, 9:17] = np.inf
        planes[4, 0, 17:25] = np.nan
        _run_vs_oracle(amd, 4, 260, planes, steps=3, budget=1.0e9, bid_lo=0.01, bid_hi=1.5)
    elif case == "drift_autoreset":
        planes = H.implicit_params(5, 520, seed=38, mean_volume=16, cvr=0.1, no_vol_prob=0.5)
        _run_vs_oracle(amd, 5, 520, planes, steps=7, budget=1e9, drift=True, check_params=True, max_days=3, loss_threshold=30.0, auto_reset=True)
    else:
        planes = H.implicit_params(6, 300, seed=39, mean_volume=16, cvr=0.3, no_vol_prob=0.4)
        n = _run_vs_oracle(amd, 6, 300, planes, steps=4, budget=6.0, bid_lo=0.5, bid_hi=1.2)
        assert n > 0


# ------------------------------------------------------------------ PHILOX mode, budget binding -> exact pass
@pytest.mark.parametrize("budget", [300.0, 25.0, 0.5, 0.01])
def test_binding_budget_matches_oracle(amd, budget):
    N, K = 6, 48
    planes = H.implicit_params(N, K, seed=11)
    n = _run_vs_oracle(amd, N, K, planes, steps=3, budget=budget, bid_lo=0.5, bid_hi=1.2)
    assert n > 0          # the exact pass really ran


@pytest.mark.parametrize("case", ["dense_256", "narrow_tiles", "budget_ground_down", "tiny_budget", "alternating", "drift_metrics", "filtered_lists",
                                  "filtered_tiny_budget", "too_many_clicks", "binds_at_once", "overflow", "off"])
def test_click_walk_matches_oracle(amd, monkeypatch, case):
    """k_step_click_walk (binding budgets, K <= 256): from the second binding day on the fast pass lists its clicked wins and
    the walk over the sorted lists replaces the row-by-row re-run.  Bit-exact against the oracle (which knows neither), with the
    counters showing that the path really ran - complete lists, lists filtered by the previous day's hints (a day of more clicks
    than one group holds), campaigns that stop (tiny budgets: the impressions after the stop are taken off again); a list that
    overflows, a day with more clicks than the lists are worth, and the path switched off give the same results through the row
    kernel."""
    N, K, vol, steps, budgets = 6, 256, 40, 5, [900.0]       # (a day costs ~1300 here, 40 % of it in sub-timestep 0)
    drift = metrics = False
    if case == "narrow_tiles": N, K, budgets = 2, 100, [330.0]
    if case == "budget_ground_down": budgets = [60.0]
    if case == "tiny_budget": budgets = [0.37]
    if case == "alternating": budgets, steps = [850.0, 850.0, 1e9, 850.0, 1e9, 1e9, 700.0, 700.0], 8
    if case == "drift_metrics": drift = metrics = True; N, K = 5, 200
    if case in ("filtered_lists", "filtered_tiny_budget"):
        vol, steps, budgets = 110, 6, [2200.0] if case == "filtered_lists" else [0.9]
        monkeypatch.setenv("ADCRAFT_CLICK_WALK_MAX", "100000")
    if case == "too_many_clicks": vol = 110
    if case == "binds_at_once": budgets = [200.0]            # in sub-timestep 0
    if case == "overflow": monkeypatch.setenv("ADCRAFT_CLICK_CAP", "16")
    if case == "off": monkeypatch.setenv("ADCRAFT_CLICK_WALK", "0")
    planes = H.implicit_params(N, K, seed=41, mean_volume=vol)
    e = amd.StepEngine(N, K, seed=9, drift_enabled=drift)
    e.set_all_params(planes)
    e.reset()
    e.walk_stats(reset=True)
    if metrics:
        e.metrics_enable(True)
        e.metrics_reset()
    o = H.mirror_oracle(e, planes, drift_on=drift)
    prof = np.zeros(K, dtype=np.int64)
    bound = 0
    for s in range(steps):
        budget = budgets[s % len(budgets)]
        bids = o.sample_bids(0.4, 1.2)
        got, ref = e.step(bids, budget), o.step(bids, budget)
        H.assert_step_equal(got, ref, implicit=True)
        bound += int((ref["cost_cents"].sum(axis=1) >= np.rint(np.float64(np.float32(budget)) * 100)).sum())
        prof += (np.rint(got["revenue"].astype(np.float64) * 100) - np.rint(got["cost"].astype(np.float64) * 100)).sum(axis=0).astype(np.int64)
    walked, overflowed, stopped, other = e.walk_stats()
    assert bound > 0
    if case in ("off", "too_many_clicks"): assert walked + overflowed + stopped + other == 0
    elif case == "overflow": assert overflowed > 0 and walked == 0
    else: assert walked > 0 and overflowed == 0 and walked + other >= bound - 2 * N       # all but each env's first binding day(s)
    if case in ("tiny_budget", "filtered_tiny_budget"): assert stopped > 0
    if metrics:
        kp, sc = e.metrics_read()
        assert np.array_equal(kp, prof)
        o.materialize_drift()
        assert np.array_equal(e.get_all_params(), o.params)
    e.close()


@pytest.mark.parametrize("case", ["lists", "few_marks_then_table", "table_only", "one_kernel"])
@pytest.mark.parametrize("budget", [700.0, 40.0, 0.4])
def test_rest_of_day_pair_matches_oracle(amd, monkeypatch, case, budget):
    """K <= 256: k_step_rest_of_day<true> hands the cells that may hold an affordable click to k_rest_walk as an ordered list
    (up to 1024 per env), beyond that as the whole table; ADCRAFT_REST_SPLIT=0 keeps both halves in one kernel.  All four give
    the oracle's days, campaigns that stop included (budget 0.4).  In k_rest_walk every lane resolves its own cell; cells of
    many auctions or clicks go through the whole wavefront (the third shape)."""
    monkeypatch.setenv("ADCRAFT_CLICK_WALK", "0")            # every binding day goes rows -> rest of day
    if case == "few_marks_then_table": monkeypatch.setenv("ADCRAFT_REST_MARKS", "3")
    if case == "table_only": monkeypatch.setenv("ADCRAFT_REST_MARKS", "0")
    monkeypatch.setenv("ADCRAFT_REST_SPLIT", "0" if case == "one_kernel" else "1")       # (the default splits from 1024 envs on)
    for N, K, seed, vol in ((5, 256, 51, 60), (3, 77, 52, 60), (2, 40, 53, 2500), (2, 300, 54, 50), (2, 1024, 55, 40)):     # (cells of 2-3 auctions; of a hundred; several keywords per lane)
        planes = H.implicit_params(N, K, seed=seed, mean_volume=vol)
        n = _run_vs_oracle(amd, N, K, planes, steps=3, budget=budget, bid_lo=0.4, bid_hi=1.2)
        assert n > 0


@pytest.mark.parametrize("drift", [False, True])
def test_rest_of_day_at_once_matches_oracle(amd, monkeypatch, drift):
    """A budget that runs out within the first cells of the day: from the next day on k_tail_or_flag parks the env for
    k_step_rest_of_day at once (hint 6) - no fast pass, no row kernel; the pass draws the volumes and applies the drift itself.
    Bit-exact against the oracle through budgets that stay small, grow (back through the row kernel), stop binding, are zero
    (the first cell is still visited, :230-233) and exhaust to the cent (campaign stop); the counter shows the path ran."""
    monkeypatch.setenv("ADCRAFT_CLICK_WALK", "0")
    monkeypatch.setenv("ADCRAFT_REST_SPLIT", "1")            # (the default splits from 1024 envs on)
    for N, K, seed, vol in ((6, 256, 61, 40), (3, 77, 62, 60)):
        planes = H.implicit_params(N, K, seed=seed, mean_volume=vol)
        e = amd.StepEngine(N, K, seed=11, drift_enabled=drift, max_days=1000, loss_threshold=1e9)
        e.set_all_params(planes)
        e.reset()
        e.direct_days(reset=True)
        o = H.mirror_oracle(e, planes, drift_on=drift, max_days=1000, loss_threshold=1e9)
        budgets = [3.0, 3.0, 3.0, 2.5, 20.0, 20.0, 400.0, 3.0, 3.0, 1e9, 3.0, 3.0, 3.0, 0.0, 0.05, 0.05, 0.05, 3.0]
        seen = []
        for budget in budgets:
            bids = o.sample_bids(0.4, 1.2)
            H.assert_step_equal(e.step(bids, budget), o.step(bids, budget), implicit=True)
            seen.append(e.direct_days())
        assert seen[1] == 0 and seen[2] > 0          # the first binding day tells the host, the second earns the hint, the third uses it
        assert seen[-1] > seen[10] > seen[3]         # ... and again after the larger budgets and the day that did not bind
        if drift:
            o.materialize_drift()
            assert np.array_equal(e.get_all_params(), o.params)
        e.close()


@pytest.mark.parametrize("path", ["rest_table", "click_lists"])
def test_lazy_allocation_failure_turns_the_path_off_not_the_step(amd, monkeypatch, path):
    """The two big buffers of the budget-exact paths are allocated when the device first asks for them (the rest-of-day table once
    a budget binds, the click lists' records once an env wants lists).  Both paths are scheduling choices, so a hipMalloc that
    fails then (ADCRAFT_FAIL_LAZY_ALLOC=1 makes it) must cost nothing but the path: every step still equals the oracle, no call
    raises, the drift and the metric sums are applied once (the step used to fail half-enqueued)."""
    monkeypatch.setenv("ADCRAFT_FAIL_LAZY_ALLOC", "1")
    if path == "rest_table":
        monkeypatch.setenv("ADCRAFT_CLICK_WALK", "0")
        monkeypatch.setenv("ADCRAFT_REST_SPLIT", "1")        # the pair of kernels: the one that needs the table
    N, K = 6, 96
    planes = H.implicit_params(N, K, seed=71, mean_volume=30)
    e = amd.StepEngine(N, K, seed=17, drift_enabled=True)
    e.metrics_enable(True)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=True)
    profit = np.zeros(K, np.int64)
    for budget in (25.0, 25.0, 25.0, 1e9, 25.0, 25.0):
        bids = o.sample_bids(0.3, 1.0)
        got, ref = e.step(bids, budget), o.step(bids, budget)
        H.assert_step_equal(got, ref, implicit=True)
        profit += (ref["revenue_cents"] - ref["cost_cents"]).sum(axis=0)
        assert e.step_kernel_name() == "k_step_implicit_fast<true>"          # never the listing variant: its records do not exist
    kp, _ = e.metrics_read()
    assert np.array_equal(kp, profit)
    o.materialize_drift()
    assert np.array_equal(e.get_all_params(), o.params)
    e.close()


def test_float_chain_by_integer_scan(amd):
    """The reference's `remaining_budget -= sum(costs)` chain (bidding_simulation.py:225) is float64 arithmetic, rounded after every
    cell, and its residue decides exact ties (B-15).  The kernels subtract 64 cell sums at a time by an integer prefix scan inside the
    running value's binade (common.inc chain_subtract_wave).  That must be THE float64 of the plain chain, bit for bit: checked on the
    device against its own one-lane chain and against numpy's, on random laws and on the cases the scan has to hand to the hardware -
    exact halves (round-half-even), results that leave the binade, operands as large as the running value, zeros, a chain that
    runs the value down to (and through) zero, negative and subnormal operands."""
    import ctypes as C
    from adcraft_amd import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(77)

    def run(r0, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(2, np.float64)
        _ffi.check(L.adc_debug_chain_device(0, float(r0), x.size, x.ctypes.data, out.ctypes.data))
        ref = np.float64(r0)
        for v in x:
            ref = ref - v                      # float64, one rounding per step
        assert out[0].tobytes() == out[1].tobytes() == np.float64(ref).tobytes(), (r0, out, ref)
        return out[0]

    # the budgets and cell sums the engine meets: cents / 100 in float64, sums of a few of them
    for trial in range(60):
        n = int(rng.integers(1, 700))
        cents = rng.integers(0, 400, (n, 4)) * (rng.random((n, 4)) < 0.4)
        x = np.zeros(n)
        for j in range(4):
            x = x + cents[:, j] / 100.0        # left to right, as a cell's costs are summed
        run(float(rng.choice([1000.0, 10.0, 37.5, 123456.78, 0.5 + x.sum(), x.sum()])), x)
    # exact halves of the running value's ulp, on both parities, and just beside them
    for r0 in (600.0, 600.0 + 2.0 ** -43, 1.0, 2.0 - 2.0 ** -52, 1023.9999999):
        u = np.spacing(np.float64(r0))
        base = rng.integers(1, 1 << 20, 300).astype(np.float64) * u
        for off in (0.5 * u, 0.5 * u + np.spacing(0.5 * u), 0.5 * u - np.spacing(0.25 * u), 0.25 * u, 0.75 * u, 0.0):
            run(r0, base + off)
        run(r0, np.full(200, 0.5 * u))                 # nothing but ties
        run(r0, np.full(200, 1.5 * u))
    # binade exits: a value ground down over many octaves, operands comparable to it, the last step to exactly zero and below
    run(1024.0, np.array([512.0, 256.0, 128.0, 64.0, 32.0, 16.0, 8.0, 4.0, 2.0, 1.0, 0.5, 0.25, 0.25, 0.125]))
    run(1000.0, np.concatenate([np.full(99, 10.0), [9.99, 0.01], [0.01] * 3]))
    run(5.0, rng.random(500) * 0.02)
    x = rng.random(640) * 3.0
    run(float(np.float64(x[:320].sum())), x)                        # crosses zero somewhere in the middle
    run(1.0e15, rng.random(300) * 1.0e13)
    run(0.07, np.array([0.01] * 9))
    # operands the scan does not take itself
    run(100.0, np.array([1.0, -2.5, 0.0, -0.0, 5e-324, 1e-310, 3.0, -1e-3] * 20))
    run(100.0, np.zeros(130))
    run(-3.0, rng.random(70))
    run(0.0, rng.random(70))
    # random magnitudes across the whole exponent range below the running value
    for trial in range(40):
        r0 = float(2.0 ** rng.uniform(-20, 40))
        x = r0 * 2.0 ** rng.uniform(-70, -1, 500) * (rng.random(500) < 0.8)
        run(r0, x)


@pytest.mark.parametrize("groups", ["2", "3", "4"])
def test_env_groups_on_their_own_streams_change_nothing(amd, monkeypatch, groups):
    """From 2048 envs on an IMPLICIT step runs as 2 or 4 env GROUPS - contiguous env ranges, each with its own view of the engine's
    arrays, its own lists and counters, its own stream - so that one group's latency-bound kernels run under another's keyword-parallel
    pass.  Scheduling only: forced here on a handful of envs (ADCRAFT_STREAM_GROUPS, uneven group sizes included), every path a step
    can take gives the oracle's results - the budget-free pass (both kernels), budgets that bind (row kernel, rest-of-day pair,
    at-once parking, click lists), drift, metric sums, auto-reset, host steps interleaved with device-resident ones, a profiled
    stretch (one group while events bracket the kernels) in the middle."""
    monkeypatch.setenv("ADCRAFT_STREAM_GROUPS", groups)
    monkeypatch.setenv("ADCRAFT_REST_SPLIT", "1")
    for case in ("dense", "sparse", "lists", "rest_pair", "at_once"):
        N, K = (7, 256) if case != "sparse" else (5, 300)
        if case == "lists":
            monkeypatch.delenv("ADCRAFT_CLICK_WALK", raising=False)
        elif case in ("rest_pair", "at_once"):
            monkeypatch.setenv("ADCRAFT_CLICK_WALK", "0")
        if case == "sparse":
            monkeypatch.setenv("ADCRAFT_FAST_VARIANT", "2")
            monkeypatch.setenv("ADCRAFT_FAST_TILE_KW", "256")
            planes = H.implicit_params(N, K, seed=91, mean_volume=16, cvr=0.1, no_vol_prob=0.5)
        else:
            monkeypatch.delenv("ADCRAFT_FAST_VARIANT", raising=False)
            monkeypatch.delenv("ADCRAFT_FAST_TILE_KW", raising=False)
            planes = H.implicit_params(N, K, seed=92, mean_volume=40)
        budgets = {"dense": [1e9, 1e9, 900.0, 1e9], "sparse": [1e9, 3.0, 3.0, 1e9], "lists": [900.0] * 5,
                   "rest_pair": [700.0, 700.0, 40.0, 700.0, 1e9, 700.0], "at_once": [3.0, 3.0, 3.0, 3.0, 20.0, 3.0, 3.0]}[case]
        e = amd.StepEngine(N, K, seed=23, drift_enabled=True, max_days=4, loss_threshold=1e9, auto_reset=True)
        e.set_all_params(planes)
        e.reset()
        e.metrics_enable(True)
        e.metrics_reset()
        o = H.mirror_oracle(e, planes, drift_on=True, max_days=4, loss_threshold=1e9, auto_reset=True)
        profit = np.zeros(K, np.int64)
        for i, budget in enumerate(budgets):
            bids = o.sample_bids(0.4, 1.2)
            if i == 2:
                e.profile_enable(True)
            if i % 2 == 0:
                got = e.step(bids, budget)
            else:                                   # device-resident: the engine's own action stream (the oracle's sample_bids), fetched afterwards
                e.sample_actions(0.4, 1.2, budget)
                e.step_device()
                got = e.fetch()
            if i == 2:
                e.profile_enable(False)
            ref = o.step(bids, budget)
            H.assert_step_equal(got, ref, implicit=True)
            profit += (ref["revenue_cents"] - ref["cost_cents"]).sum(axis=0)
        kp, sc = e.metrics_read()
        assert np.array_equal(kp, profit) and sc[1] == len(budgets) * N
        o.materialize_drift()
        assert np.array_equal(e.get_all_params(), o.params)
        e.close()


@pytest.mark.parametrize("model,K,groups", [(1, 96, 3), (1, 700, 4), (2, 300, 2), (2, 64, 4)])
def test_env_groups_of_the_float_money_models(amd, model, K, groups):
    """the same for ExplicitKeyword and the default ImplicitKeyword (their steps are chains of one-workgroup-per-env kernels): forced
    on a handful of envs, uneven groups, budgets that bind early, late and not at all, drift, metric sums, host steps and device-
    resident ones, a profiled step in between"""
    N = 7
    if model == 1:
        planes = H.explicit_params(N, K, seed=61)
        extra = {}
        lo, hi, budgets = 0.05, 2.0, (K * 1.5, K * 1.5, 1e9, K * 0.1, 1e9, K * 4.0)
    else:
        rng = np.random.default_rng(62)
        planes = np.stack([rng.integers(0, 60, (N, K)), rng.random((N, K)) * 6, rng.uniform(0.0, 0.3, (N, K)), rng.uniform(0.05, 0.15, (N, K)),
                           rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.3, 1.5, (N, K)),
                           rng.uniform(0.02, 0.3, (N, K))]).astype(np.float32)
        extra = dict(max_bidders=30, participation_rate=0.6, num_winners=1)
        lo, hi, budgets = 0.05, 0.5, (K * 0.05, K * 0.05, 1e9, K * 0.004, 1e9, K * 0.2)
    e = amd.StepEngine(N, K, model=model, seed=19, drift_enabled=True)
    if model == 2:
        e.set_general_model(30, 0.6, 1)
    e.set_env_groups(groups)
    e.set_all_params(planes)
    e.reset()
    e.metrics_enable(True)
    e.metrics_reset()
    o = H.mirror_oracle(e, planes, drift_on=True, **extra)
    bound = 0
    for i, b in enumerate(budgets):
        bids = o.sample_bids(lo, hi)
        if i == 3:
            e.profile_enable(True)
        if i % 2 == 0:
            got = e.step(bids, b)
        else:
            e.sample_actions(lo, hi, b)
            e.step_device()
            got = e.fetch()
        assert e.env_groups() == (1 if i == 3 else groups)
        if i == 3:
            e.profile_enable(False)
        ref = o.step(bids, b)
        H.assert_step_equal(got, ref, implicit=False)
        bound += int((ref["cost"].astype(np.float64).sum(axis=1) >= 0.98 * b).sum())
    assert bound >= N          # the budgets really bound
    _, sc = e.metrics_read()
    assert sc[1] == len(budgets) * N
    o.materialize_drift()
    assert np.array_equal(e.get_all_params(), o.params)
    e.close()


@pytest.mark.parametrize("queues", ["1", "2"])
def test_env_groups_with_fewer_hardware_queues(queues):
    """the groups' streams are probed: with the runtime limited to one or two hardware queues (GPU_MAX_HW_QUEUES, read when the process
    initialises HIP - hence a child process) the engine finds one or two streams that overlap and runs that many groups, not four on
    shared queues; results as ever"""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.')\n"
        "from tests import helpers as H\n"
        "from adcraft_amd.engine import StepEngine\n"
        "N, K = 9, 64\n"
        "planes = H.implicit_params(N, K, seed=97, mean_volume=30)\n"
        "out = []\n"
        "for groups in (1, 4):\n"
        "    e = StepEngine(N, K, seed=43, drift_enabled=True)\n"
        "    e.set_env_groups(groups); e.set_all_params(planes); e.reset()\n"
        "    e.sample_actions(0.3, 1.0, 25.0)\n"
        "    for _ in range(5): e.step_device()\n"
        "    o = e.fetch(); out.append((o, e.get_all_params(), e.env_groups())); e.close()\n"
        "(a, ap, ag), (b, bp, bg) = out\n"
        "assert all(np.array_equal(a[k], b[k]) for k in a) and np.array_equal(ap, bp)\n"
        "print('GROUPS', ag, bg)\n")
    env = dict(os.environ, GPU_MAX_HW_QUEUES=queues)
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [ln for ln in r.stdout.splitlines() if ln.startswith("GROUPS")][-1].split()
    assert got[1] == "1" and 1 <= int(got[2]) <= int(queues), r.stdout


def test_env_groups_at_the_size_the_engine_chooses_them(amd):
    """2048 envs: the engine's own choice against the oracle.  Groups are for CHAINED device-resident steps only (a step that follows a
    step with no other call in between: 4 groups for a dense keyword set, 2 for a sparse one); a step behind any other call - a host
    step, new actions, a fetch - and a profiled step run as one group.  Budget-free and binding steps, then a fetch"""
    N, K = 2048, 32
    for mean_volume, want in ((40, 4), (8, 2)):
        planes = H.implicit_params(N, K, seed=93, mean_volume=mean_volume)
        e = amd.StepEngine(N, K, seed=29, drift_enabled=True)
        e.set_all_params(planes)
        e.reset()
        o = H.mirror_oracle(e, planes, drift_on=True, threads=8)
        for budget in (1e9, 12.0, 12.0):
            bids = o.sample_bids(0.3, 1.0)
            H.assert_step_equal(e.step(bids, budget), o.step(bids, budget), implicit=True)
            assert e.env_groups() == 1                   # (a host step is joined at once: one group)
        for budget in (12.0, 1e9, 12.0):        # device-resident, back to back: the groups run ahead of each other
            e.sample_actions(0.3, 1.0, budget)
            bids = o.sample_bids(0.3, 1.0)
            for _ in range(4):                      # (the same bids four days running: the first step behind sample_actions is one group, the chain behind it is grouped)
                ref = o.step(bids, budget)
                e.step_device()
            assert e.env_groups() == want
            H.assert_step_equal(e.fetch(), ref, implicit=True)
            e.step_device()
            assert e.env_groups() == 1                   # (behind the fetch)
            o.step(bids, budget)
        e.profile_enable(True)
        bids = o.sample_bids(0.3, 1.0)
        H.assert_step_equal(e.step(bids, 12.0), o.step(bids, 12.0), implicit=True)
        assert e.env_groups() == 1
        e.profile_enable(False)
        o.materialize_drift()
        assert np.array_equal(e.get_all_params(), o.params)
        e.close()


def test_mixed_binding_and_not(amd):
    """some envs hit the budget, others do not, in the same launch"""
    N, K = 8, 64
    planes = H.implicit_params(N, K, seed=12)
    planes[0, ::2] = 8.0          # low-volume envs never reach the budget
    planes[1, ::2] = 1.0
    n = _run_vs_oracle(amd, N, K, planes, steps=3, budget=60.0)
    assert 0 < n < 3 * N


def test_episode_tail_termination_truncation_autoreset(amd):
    N, K = 4, 32
    planes = H.implicit_params(N, K, seed=13, cvr=0.1)       # loses money
    _run_vs_oracle(amd, N, K, planes, steps=9, budget=1e9, max_days=4, loss_threshold=30.0, auto_reset=True,
                   bid_lo=0.8, bid_hi=1.4)


def test_drift_matches_oracle(amd):
    N, K = 5, 200
    planes = H.implicit_params(N, K, seed=14)
    _run_vs_oracle(amd, N, K, planes, steps=5, budget=1e9, drift=True, check_params=True)
    _run_vs_oracle(amd, N, K, planes, steps=4, budget=40.0, drift=True, check_params=True)


# ------------------------------------------------------------------ EXPLICIT model (default constructor)
@pytest.mark.parametrize("budget", [1000.0, 6.0])
def test_explicit_matches_oracle(amd, budget):
    N, K = 3, 64
    planes = H.explicit_params(N, K, seed=15)
    _run_vs_oracle(amd, N, K, planes, steps=3, budget=budget, model=1, bid_lo=0.05, bid_hi=2.0)


@pytest.mark.parametrize("model,K", [(1, 300), (1, 512), (1, 700), (1, 1024), (2, 300), (2, 1000)])
def test_float_day_kernel_beyond_256_keywords(amd, model, K):
    """binding budgets for the float-money models at 256 < K <= 1024: the day kernel with 512 / 1024 lanes (lane = keyword) instead
    of the one-wavefront walkers - bit-exact against the oracle with drift, budgets that bind early, late and not at all, and the
    hint that skips the keyword-parallel pass the day after a binding one"""
    N = 3
    if model == 1:
        planes = H.explicit_params(N, K, seed=51)
        extra = {}
        lo, hi, budgets = 0.05, 2.0, (K * 1.5, K * 1.5, 1e9, K * 0.1, 1e9, K * 4.0)
    else:
        rng = np.random.default_rng(52)
        planes = np.stack([rng.integers(0, 60, (N, K)), rng.random((N, K)) * 6, rng.uniform(0.0, 0.3, (N, K)), rng.uniform(0.05, 0.15, (N, K)),
                           rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.3, 1.5, (N, K)),
                           rng.uniform(0.02, 0.3, (N, K))]).astype(np.float32)
        extra = dict(max_bidders=30, participation_rate=0.6, num_winners=1)
        lo, hi, budgets = 0.05, 0.5, (K * 0.05, K * 0.05, 1e9, K * 0.004, 1e9, K * 0.2)
    e = amd.StepEngine(N, K, model=model, seed=17, drift_enabled=True)
    if model == 2:
        e.set_general_model(30, 0.6, 1)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=True, **extra)
    bound = 0
    for b in budgets:
        bids = o.sample_bids(lo, hi)
        got, ref = e.step(bids, b), o.step(bids, b)
        H.assert_step_equal(got, ref, implicit=False)
        bound += int((ref["cost"].astype(np.float64).sum(axis=1) >= 0.98 * b).sum())
    assert bound >= N          # the budgets really bound
    o.materialize_drift()
    assert np.array_equal(e.get_all_params(), o.params)
    e.close()


@pytest.mark.parametrize("case", ["full_width", "pool_exhausted", "huge_cells", "no_budget", "narrow", "drift_with_hint"])
def test_explicit_day_kernel_and_its_fallbacks(amd, case):
    """k_step_explicit_day (K <= 256: lane = keyword, one row of click lists in LDS) against the oracle, and every way out
    of it: the shared overflow pool exhausted (-> k_step_explicit_rows), cells of more than 255 auctions (-> the serial
    walker), a day that starts without budget; plus the scheduling hint that skips the fast pass on the following days
    (drift then has to be applied by the day kernel)"""
    steps, drift, budget = 4, False, 40.0
    if case == "full_width":
        N, K = 5, 256
        planes = H.explicit_params(N, K, seed=31)
    elif case == "pool_exhausted":            # 23 auctions, all in the day's first cell, nearly all shown and clicked: ~5000 per row
        N, K = 2, 256
        planes = H.explicit_params(N, K, seed=32)
        planes[0], planes[1], planes[4] = 23.0, 0.0, 0.97
        planes[2], planes[3] = 0.0, 25.0                                    # steep sigmoid at 0: every bid shows
        budget = 1500.0
    elif case == "huge_cells":
        N, K = 2, 6
        planes = H.explicit_params(N, K, seed=33)
        planes[0], planes[1] = 9000.0, 100.0
        budget, steps = 900.0, 2
    elif case == "no_budget":
        N, K = 3, 40
        planes = H.explicit_params(N, K, seed=34)
        budget = 0.0
    elif case == "narrow":
        N, K = 6, 7
        planes = H.explicit_params(N, K, seed=35)
        budget = 3.0
    else:
        N, K = 4, 96
        planes = H.explicit_params(N, K, seed=36)
        steps, drift, budget = 6, True, 25.0
    n_bound = _run_vs_oracle(amd, N, K, planes, steps=steps, budget=budget, model=1, drift=drift, check_params=drift, bid_lo=0.05, bid_hi=2.0)
    if case != "no_budget":
        assert n_bound >= 0


def test_explicit_drift(amd):
    planes = H.explicit_params(2, 10, seed=16)
    _run_vs_oracle(amd, 2, 10, planes, steps=3, budget=1000.0, model=1, drift=True, check_params=True)


# ------------------------------------------------------------------ TAPE mode: the reference's own traces
def test_g3_implicit_replay_on_gpu(amd, golden):
    for t in golden("g3_implicit_replay.json")["traces"]:
        K = t["K"]
        e = amd.StepEngine(1, K, seed=1)
        e.reset()
        tp = t["tape"]
        tape = amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), bid_cents=tp["bid"], click=tp["click"],
                              conv=tp["conv"], rev_cents=tp["rev"])
        out = e.step_replay(np.array(t["bids"], np.float32), t["budget"], tape)
        ref = t["out"]
        assert out["impressions"][0].tolist() == ref["impressions"]
        assert out["buyside_clicks"][0].tolist() == ref["buyside_clicks"]
        assert out["sellside_conversions"][0].tolist() == ref["sellside_conversions"]
        assert np.array_equal(out["cost"][0], np.array(ref["cost"]).astype(np.float32))
        assert np.array_equal(out["revenue"][0], np.array(ref["revenue"]).astype(np.float32))
        np.testing.assert_allclose(out["reward"][0], sum(ref["profit"]), rtol=0, atol=1e-9)
        assert (tape.end["bid"][0], tape.end["click"][0], tape.end["conv"][0], tape.end["rev"][0]) == \
            (len(tp["bid"]), len(tp["click"]), len(tp["conv"]), len(tp["rev"]))
        e.close()


def test_g3_explicit_replay_on_gpu(amd, golden):
    for t in golden("g3_explicit_replay.json")["traces"]:
        K = t["K"]
        e = amd.StepEngine(1, K, model=1, seed=1)
        e.reset()
        tp = t["tape"]
        tape = amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), click=tp["click"], conv=tp["conv"],
                              rev_cents=tp["rev"], x_impressions=tp["impressions"], x_cost=tp["cost"])
        out = e.step_replay(np.array(t["bids"], np.float32), t["budget"], tape)
        ref = t["out"]
        assert out["impressions"][0].tolist() == ref["impressions"]
        assert out["buyside_clicks"][0].tolist() == ref["buyside_clicks"]
        assert out["sellside_conversions"][0].tolist() == ref["sellside_conversions"]
        assert np.array_equal(out["cost"][0], np.array(ref["cost"]).astype(np.float32))     # f64 sum in reference order
        np.testing.assert_allclose(out["reward"][0], sum(ref["profit"]), rtol=0, atol=1e-9)
        assert tape.end["xcost"][0] == len(tp["cost"]) and tape.end["click"][0] == len(tp["click"])
        e.close()


def _implicit_planes(kp):
    return np.array([[p["vol_mean"] for p in kp], [p["vol_std"] for p in kp], [p["loc"] for p in kp], [p["scale"] for p in kp],
                     [p["bctr"] for p in kp], [p["sctr"] for p in kp], [p["rev_mean"] for p in kp], [p["rev_std"] for p in kp]],
                    np.float32).reshape(8, 1, len(kp))


def test_g3_outcome_lists_on_gpu_element_for_element(amd, golden):
    """f4, second half: the per-click lists of info["bidding_outcomes"] against the reference's own BiddingOutcomes
    (bidding_simulation.py:10-38,124-147, recorded by tools/gen_golden.py): step_replay on the reference's tape, then the
    read-only walk of the same tape (adc_engine_outcomes_replay_tape) and the facade's combination of its click records -
    costs, revenues, revenues_per_cost element for element and in order, impression_share the very same float64 (B-11's lossy
    denominator included), per-keyword profit within 1e-9; and the formatted string (src/lib.rs:251-275) parsed back."""
    import ast
    from adcraft_amd.gymnasium_kw_env import BiddingSimulation, combined_outcomes
    n_lossy = n_traces = 0
    for model, name in ((0, "g3_implicit_replay.json"), (1, "g3_explicit_replay.json")):
        for t in golden(name)["traces"]:
            K, tp = t["K"], t["tape"]
            e = amd.StepEngine(1, K, model=model, seed=1)
            if model == 0:
                e.set_all_params(_implicit_planes(t["keyword_params"]))       # (a tape replay reads none of them; the walk must not either)
            e.reset()

            def tape():
                if model == 0:
                    return amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), bid_cents=tp["bid"], click=tp["click"], conv=tp["conv"],
                                          rev_cents=tp["rev"])
                return amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"],
                                      x_impressions=tp["impressions"], x_cost=tp["cost"])
            bids = np.array(t["bids"], np.float32)
            out = e.step_replay(bids, t["budget"], tape())
            before = e.get_episode_state()
            clicks = e.outcomes_replay(0, bids, t["budget"], tape=tape())
            assert e.get_episode_state()[0].tolist() == before[0].tolist() and np.array_equal(e.get_episode_state()[1], before[1])   # read-only
            obs = {k: out[k][0] for k in ("impressions", "buyside_clicks", "sellside_conversions", "cost", "revenue")}
            rows = combined_outcomes(t["bids"], obs, clicks)
            ref = t["out"]
            got = {f: [r[f] for r in rows] for f in ("costs", "revenues", "revenues_per_cost", "impression_share", "profit")}
            H.assert_outcome_lists(got, ref, K)
            assert [r["buyside_clicks"] for r in rows] == ref["buyside_clicks"] == [len(c) for c in ref["costs"]]
            # (c) the string the facade formats, parsed back
            parsed = ast.literal_eval(BiddingSimulation._repr_outcomes(t["bids"], obs, clicks))
            assert [list(r) for r in parsed][0] == ["bid", "impressions", "impression_share", "buyside_clicks", "costs", "sellside_conversions",
                                                    "revenues", "revenues_per_cost", "profit"]
            for k, r in enumerate(parsed):
                assert r["bid"] == t["bids"][k] and r["impressions"] == ref["impressions"][k] and r["impression_share"] == ref["impression_share"][k]
                assert r["costs"] == ref["costs"][k] and r["revenues"] == ref["revenues"][k] and r["revenues_per_cost"] == ref["revenues_per_cost"][k]
                assert r["sellside_conversions"] == ref["sellside_conversions"][k] and abs(r["profit"] - ref["profit"][k]) < 1e-9
            n_lossy += sum(1 for k in range(K) if t["volumes"][k] > 0 and ref["impressions"][k] > 0
                           and ref["impression_share"][k] != ref["impressions"][k] / t["volumes"][k])
            n_traces += 1
            e.close()
    assert n_traces >= 19 and n_lossy > 10


def test_g8_outcome_lists_on_gpu_every_step(amd, golden):
    """the same through whole step() episodes (what gymnasium_kw_env.py:249 hands to repr_outcomes_py), the tape cut per step"""
    from adcraft_amd.gymnasium_kw_env import combined_outcomes
    for ep in golden("g8_env_episodes.json")["episodes"]:
        K, tp = ep["K"], ep["tape"]
        e = amd.StepEngine(1, K, seed=1, max_days=ep["max_days"], loss_threshold=ep["loss_threshold"])
        e.reset()
        for st in ep["steps"]:
            sl = st["tape_slices"]

            def tape():
                return amd.ReplayTape(1, np.array(st["volumes"]).reshape(1, K), bid_cents=tp["bid"], click=tp["click"], conv=tp["conv"],
                                      rev_cents=tp["rev"], offsets={k: [sl[k][0]] for k in ("bid", "click", "conv", "rev")})
            bids = np.array(st["bids"], np.float32)
            out = e.step_replay(bids, st["budget"], tape())
            clicks = e.outcomes_replay(0, bids, st["budget"], tape=tape())
            obs = {k: out[k][0] for k in ("impressions", "buyside_clicks", "sellside_conversions", "cost", "revenue")}
            rows = combined_outcomes(st["bids"], obs, clicks)
            H.assert_outcome_lists({f: [r[f] for r in rows] for f in ("costs", "revenues", "revenues_per_cost", "impression_share", "profit")},
                                   st["outcomes"], K)
        e.close()


def test_g12_general_implicit_replay_on_gpu(amd, golden):
    """G12: the reference's default ImplicitKeyword (Binomial bidders per call, raw Laplace bids, literal top-(w+n) clearing)
    replayed through k_step_exact<IMPLICIT_GENERAL, TAPE>: integers exact, float64 cost sums bit-identical, cursors at the ends"""
    for t in golden("g12_implicit_general_replay.json")["traces"]:
        K = t["K"]
        e = amd.StepEngine(1, K, model=2, seed=1)
        e.set_general_model(t["max_bidders"], t["participation_rate"], 1)
        kp = t["keyword_params"]
        planes = np.zeros((8, 1, K), np.float32)
        for i, name in ((2, "bid_loc"), (3, "bid_scale"), (4, "bctr"), (5, "sctr"), (6, "rev_mean"), (7, "rev_std")):
            planes[i, 0] = [p[name] for p in kp]
        e.set_all_params(planes)
        e.reset()
        tp = t["tape"]
        tape = amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"],
                              x_impressions=tp["bidders"], x_cost=tp["bids"])
        out = e.step_replay(np.array(t["bids"], np.float32), t["budget"], tape)
        ref = t["out"]
        assert out["impressions"][0].tolist() == ref["impressions"]
        assert out["buyside_clicks"][0].tolist() == ref["buyside_clicks"]
        assert out["sellside_conversions"][0].tolist() == ref["sellside_conversions"]
        assert np.array_equal(out["cost"][0], np.array(ref["cost"]).astype(np.float32))     # f64 sum in reference order
        np.testing.assert_allclose(out["reward"][0], sum(ref["profit"]), rtol=0, atol=1e-9)
        assert tape.end["ximp"][0] == len(tp["bidders"]) and tape.end["xcost"][0] == len(tp["bids"])
        assert tape.end["click"][0] == len(tp["click"]) and tape.end["conv"][0] == len(tp["conv"]) and tape.end["rev"][0] == len(tp["rev"])
        e.close()


def test_g12_outcome_lists_on_gpu_element_for_element(amd, golden):
    """the default ImplicitKeyword's per-click lists against the reference's own combined BiddingOutcomes (G12, recorded by
    tools/gen_golden_general.py): step_replay on the reference's tape, the read-only walk of the same tape
    (adc_engine_outcomes_replay_tape, model 2) and the facade's combination - float64 prices and revenues element for element
    and in order, impression_share the very float64, profit within 1e-9"""
    from adcraft_amd.gymnasium_kw_env import combined_outcomes
    n = 0
    for t in golden("g12_implicit_general_replay.json")["traces"]:
        K, tp = t["K"], t["tape"]
        e = amd.StepEngine(1, K, model=2, seed=1)
        e.set_general_model(t["max_bidders"], t["participation_rate"], 1)
        planes = np.zeros((8, 1, K), np.float32)
        for i, name in ((2, "bid_loc"), (3, "bid_scale"), (4, "bctr"), (5, "sctr"), (6, "rev_mean"), (7, "rev_std")):
            planes[i, 0] = [p[name] for p in t["keyword_params"]]
        e.set_all_params(planes)
        e.reset()

        def tape():
            return amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"],
                                  x_impressions=tp["bidders"], x_cost=tp["bids"])
        bids = np.array(t["bids"], np.float32)
        out = e.step_replay(bids, t["budget"], tape())
        clicks = e.outcomes_replay(0, bids, t["budget"], tape=tape())
        obs = {k: out[k][0] for k in ("impressions", "buyside_clicks", "sellside_conversions", "cost", "revenue")}
        rows = combined_outcomes(t["bids"], obs, clicks)
        H.assert_outcome_lists({f: [r[f] for r in rows] for f in ("costs", "revenues", "revenues_per_cost", "impression_share", "profit")},
                               t["out"], K)
        assert [r["buyside_clicks"] for r in rows] == t["out"]["buyside_clicks"] == [len(c) for c in t["out"]["costs"]]
        n += sum(t["out"]["buyside_clicks"])
        e.close()
    assert n > 1000


def _model_case(model, rng, N, K):
    """keyword planes, bid range, (ample, binding) budgets and engine / oracle settings of a small case per model"""
    if model == 0:
        return H.implicit_params(N, K, seed=77, mean_volume=40), (0.3, 1.0), (1.0e9, 30.0), {}
    if model == 1:
        return H.explicit_params(N, K, seed=78), (0.3, 1.0), (1.0e9, 25.0), {}
    planes = np.stack([rng.integers(0, 90, (N, K)), rng.random((N, K)) * 6, rng.uniform(0.0, 0.3, (N, K)), rng.uniform(0.05, 0.15, (N, K)),
                       rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.3, 1.5, (N, K)),
                       rng.uniform(0.02, 0.3, (N, K))]).astype(np.float32)
    return planes, (0.05, 0.5), (1.0e9, 3.0), dict(max_bidders=30, participation_rate=0.6, num_winners=1)


@pytest.mark.parametrize("model", [0, 1, 2])
@pytest.mark.parametrize("drift", [False, True])
@pytest.mark.parametrize("binding", [False, True])
def test_production_outcomes_replay_matches_oracle_click_for_click(amd, model, drift, binding):
    """info["bidding_outcomes"] as production builds it: adc_engine_outcomes_replay on the engine's OWN stream (the PHILOX
    instantiation of the read-only walk, addressed at tick - steps_back) against the oracle's orc_step_outcomes(tape = NULL) - every
    paid click's keyword, sub-timestep, cost and revenue in the reference's order (bidding_simulation.py:97-115,216-233), and the
    combined costs / revenues / revenues_per_cost / impression_share / profit per keyword (:124-147); for every env of the
    engine, with and without drift, binding and ample budgets, the last step and (drift off) the one before it."""
    from adcraft_amd.gymnasium_kw_env import combined_outcomes
    N, K = 3, 40
    rng = np.random.default_rng(100 + model)
    planes, (blo, bhi), budgets, general = _model_case(model, rng, N, K)
    budget = budgets[1] if binding else budgets[0]
    e = amd.StepEngine(N, K, model=model, seed=21 + model, drift_enabled=drift)
    if model == 2:
        e.set_general_model(general["max_bidders"], general["participation_rate"], general["num_winners"])
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=drift, **general)
    n_clicks = n_bound = 0
    history = []
    for step in range(4):
        bids = o.sample_bids(blo, bhi)
        got = e.step(bids, budget)
        ref, lists = o.step_outcomes(bids, budget)
        H.assert_step_equal(got, ref, implicit=(model == 0))
        history.append((bids, got, lists))
        for back in ((1,) if drift else (1, 2)):
            if back > len(history):
                continue
            b_bids, b_got, b_lists = history[-back]
            for env in range(N):
                before = e.get_episode_state()
                clicks = e.outcomes_replay(env, b_bids[env], budget, steps_back=back)
                after = e.get_episode_state()
                assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])        # read-only
                sel = b_lists["env"] == env
                assert np.array_equal(clicks["keyword"], b_lists["keyword"][sel])
                assert np.array_equal(clicks["timestep"], b_lists["timestep"][sel])
                assert np.array_equal(clicks["cost"], b_lists["cost"][sel])               # float64 dollars, bit for bit
                assert np.array_equal(clicks["revenue"], b_lists["revenue"][sel])
                obs = {k: b_got[k][env] for k in ("impressions", "buyside_clicks", "sellside_conversions", "cost", "revenue")}
                rows = combined_outcomes(b_bids[env].tolist(), obs, clicks)
                want = dict(costs=b_lists["costs"][env], revenues=b_lists["revenues"][env], revenues_per_cost=b_lists["revenues_per_cost"][env],
                            impression_share=b_lists["impression_share"][env].tolist(), profit=b_lists["profit"][env].tolist())
                H.assert_outcome_lists({f: [r[f] for r in rows] for f in ("costs", "revenues", "revenues_per_cost", "impression_share", "profit")},
                                       want, K)
                assert [r["buyside_clicks"] for r in rows] == [len(c) for c in want["costs"]]
                if back == 1:
                    n_clicks += int(sel.sum())
        spent = ref["cost_cents"].sum(axis=1) / 100.0 if model == 0 else ref["cost"].sum(axis=1)
        n_bound += int((spent > 0.8 * budget).sum())
    assert n_clicks > 50
    assert (n_bound >= N) if binding else (n_bound == 0)
    if drift:
        o.materialize_drift()
        assert np.array_equal(e.get_all_params(), o.params)
    e.close()


@pytest.mark.parametrize("small", ["0", "1"])
@pytest.mark.parametrize("budget,winners,pool", [(1e9, 1, (30, 0.6)), (3.0, 1, (30, 0.6)), (1e9, 2, (30, 0.6)), (1e9, 1, (3, 0.4)), (2.0, 2, (70, 0.5)),
                                                 (1e9, 1, (120, 0.97))])
def test_general_implicit_matches_oracle(amd, monkeypatch, budget, winners, pool, small):
    """the engine's own stream for the default ImplicitKeyword model == the C oracle, bit for bit (incl. a binding budget,
    two winning placements, pools smaller than w + n - zero padding - and larger than a wavefront, a pool whose bidder count
    goes by coins), with drift; through both keyword-parallel passes: a lane per keyword (k_step_general_fast) and a wavefront
    per keyword for a handful of envs (k_step_general_small)"""
    monkeypatch.setenv("ADCRAFT_GENERAL_SMALL", small)
    N, K = 3, 45
    rng = np.random.default_rng(8)
    planes = np.stack([rng.integers(0, 90, (N, K)), rng.random((N, K)) * 6, rng.uniform(0.0, 0.3, (N, K)), rng.uniform(0.05, 0.15, (N, K)),
                       rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.3, 1.5, (N, K)),
                       rng.uniform(0.02, 0.3, (N, K))]).astype(np.float32)
    e = amd.StepEngine(N, K, model=2, seed=13, drift_enabled=True)
    e.set_general_model(pool[0], pool[1], winners)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=True, max_bidders=pool[0], participation_rate=pool[1], num_winners=winners)
    # the budget alternates with an ample one: the keyword-parallel pass, the reference-order walker, and the hint that skips the
    # former after a day the budget bound (the walker then applies the drift) all take their turns
    for b in (budget, budget, 1e9, budget, 1e9):
        bids = o.sample_bids(0.05, 0.5)
        got, ref = e.step(bids, b), o.step(bids, b)
        H.assert_step_equal(got, ref, implicit=False)
        assert e.step_kernel_name() == ("k_step_general_small" if small == "1" else "k_step_general_fast")
    assert got["impressions"].sum() > 0
    o.materialize_drift()
    assert np.array_equal(e.get_all_params(), o.params)
    e.close()


@pytest.mark.parametrize("case", ["no_budget", "huge_cells", "full_width"])
def test_general_implicit_day_kernel_and_its_fallbacks(amd, case):
    """the default ImplicitKeyword under a binding budget: k_step_float_day<IMPLICIT_GENERAL> at full width, and its ways out
    to the reference-order walker (a day that starts without budget; cells of more than 255 auctions)"""
    rng = np.random.default_rng(18)
    N, K, budget, steps = (2, 256, 6.0, 3) if case == "full_width" else (3, 12, 0.0, 2) if case == "no_budget" else (2, 5, 8.0, 2)
    planes = np.stack([rng.integers(0, 60, (N, K)), rng.random((N, K)) * 6, rng.uniform(0.0, 0.3, (N, K)), rng.uniform(0.05, 0.15, (N, K)),
                       rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.2, 0.9, (N, K)), rng.uniform(0.3, 1.5, (N, K)),
                       rng.uniform(0.02, 0.3, (N, K))]).astype(np.float32)
    if case == "huge_cells":
        planes[0], planes[1] = 7000.0, 50.0
    e = amd.StepEngine(N, K, model=2, seed=14, drift_enabled=True)
    e.set_general_model(9, 0.5, 1)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=True, max_bidders=9, participation_rate=0.5, num_winners=1)
    for _ in range(steps):
        bids = o.sample_bids(0.05, 0.5)
        got, ref = e.step(bids, budget), o.step(bids, budget)
        H.assert_step_equal(got, ref, implicit=False)
    o.materialize_drift()
    assert np.array_equal(e.get_all_params(), o.params)
    e.close()


def test_g8_env_episodes_on_gpu(amd, golden):
    for ep in golden("g8_env_episodes.json")["episodes"]:
        K = ep["K"]
        e = amd.StepEngine(1, K, seed=1, max_days=ep["max_days"], loss_threshold=ep["loss_threshold"])
        e.reset()
        tp = ep["tape"]
        for i, st in enumerate(ep["steps"]):
            sl = st["tape_slices"]
            tape = amd.ReplayTape(1, np.array(st["volumes"]).reshape(1, K), bid_cents=tp["bid"], click=tp["click"],
                                  conv=tp["conv"], rev_cents=tp["rev"],
                                  offsets={k: [sl[k][0]] for k in ("bid", "click", "conv", "rev")})
            out = e.step_replay(np.array(st["bids"], np.float32), st["budget"], tape)
            ob = st["obs"]
            assert out["impressions"][0].tolist() == ob["impressions"]
            assert out["buyside_clicks"][0].tolist() == ob["buyside_clicks"]
            assert out["sellside_conversions"][0].tolist() == ob["sellside_conversions"]
            assert np.array_equal(out["cost"][0], np.array(ob["cost"]).astype(np.float32))
            np.testing.assert_allclose(out["reward"][0], st["reward"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(out["cumulative_profit"][0], ob["cumulative_profit"][0], rtol=0, atol=1e-9)
            assert out["days_passed"][0] == i + 1
            assert bool(out["terminated"][0]) == st["terminated"] and bool(out["truncated"][0]) == st["truncated"]
            assert tape.end["bid"][0] == sl["bid"][1] and tape.end["rev"][0] == sl["rev"][1]
        e.close()


def test_g4_drift_replay_on_gpu(amd, golden):
    """G4: update_keywords() of the reference (gymnasium_kw_env.py:114-158) replayed through k_step_exact<., TAPE>: the tape
    carries the three uniform vectors the reference drew; the engine ends the replayed step with update_keywords() on exactly
    those coefficients.  The engine holds parameters in float32, the reference in float64: each update is one f32 fma (volume)
    or one f32 multiply + clip (rates), so after s steps the relative difference is at most about s * 2^-23.
    Tolerance: rtol 2e-6 (+ atol 2e-6 for the volume, whose increments are ~0.03 * std)."""
    for seq in golden("g4_update_keywords.json")["sequences"]:
        K = seq["K"]
        up = dict((n, v) for n, v in seq["updater_params"])
        e = amd.StepEngine(1, K, seed=1, drift_enabled=True, drift=(up["vol"], up["ctr"], up["cvr"]))
        p0 = seq["params0"]
        planes = np.array([[p[0][0] for p in p0], [p[0][1] for p in p0], [p[1] for p in p0], [1.0 / p[2] for p in p0],
                           [p[3] for p in p0], [p[4] for p in p0], [p[5] for p in p0], [p[6] for p in p0]], np.float32).reshape(8, 1, K)
        e.set_all_params(planes)
        e.reset()
        for st in seq["steps"]:
            tape = amd.ReplayTape(1, np.zeros((1, K), np.int32), drift_uniforms=np.array(st["uniforms"]).reshape(3, 1, K))
            out = e.step_replay(np.full((1, K), 0.5, np.float32), 10.0, tape)
            assert out["impressions"].sum() == 0 and out["reward"][0] == 0.0          # zero-volume day: nothing but the drift happens
            got = e.get_all_params()[:, 0]
            ref = st["params"]
            np.testing.assert_allclose(got[0], [p[0][0] for p in ref], rtol=2e-6, atol=2e-6)
            assert np.array_equal(got[1], planes[1, 0])                                 # vol std never changes (:146-149)
            np.testing.assert_allclose(got[4], st["kw_bctr"], rtol=2e-6)
            np.testing.assert_allclose(got[5], st["kw_sctr"], rtol=2e-6)
            np.testing.assert_allclose(got[4], [p[3] for p in ref], rtol=2e-6)
            np.testing.assert_allclose(got[5], [p[4] for p in ref], rtol=2e-6)
        e.close()


def test_drift_tape_matches_oracle_bitwise_and_needs_drift_enabled(amd):
    """the HIP replay of a tape with drift coefficients == the C oracle's, bit for bit (both hold float32 parameters), on days
    that also have auctions; a tape with drift coefficients on an engine without drift is refused"""
    from oracle import capi as orc
    K = 40
    planes = H.implicit_params(1, K, seed=77, mean_volume=20)
    rng = np.random.default_rng(5)
    e = amd.StepEngine(1, K, seed=3, drift_enabled=True)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes, drift_on=True)
    for _ in range(3):
        u = rng.uniform(-0.03, 0.03, (3, 1, K)).astype(np.float32)
        vol = rng.integers(0, 30, (1, K)).astype(np.int32)
        n = int(vol.sum())
        comp = rng.integers(10, 120, n).astype(np.int32)
        click = rng.integers(0, 2, n).astype(np.uint8)
        conv = rng.integers(0, 2, n).astype(np.uint8)
        rev = rng.integers(1, 300, n).astype(np.int32)
        bids = rng.uniform(0.3, 1.0, (1, K)).astype(np.float32)
        got = e.step_replay(bids, 1e9, amd.ReplayTape(1, vol, bid_cents=comp, click=click, conv=conv, rev_cents=rev, drift_uniforms=u))
        ts = orc.TapeSource(bid_cents=comp, click=click, conv=conv, rev_cents=rev)
        ts.set_volumes(vol)
        ts.set_drift_uniforms(u)
        ref = o.step(bids, 1e9, tape=ts)
        H.assert_step_equal(got, ref)
        assert np.array_equal(e.get_all_params(), o.params)          # drift applied at once on both sides, same float32 ops
        assert not o.drift_pending.any()
    assert not np.array_equal(o.params[0], planes[0])
    e.close()
    e2 = amd.StepEngine(1, 4, seed=3)          # drift not enabled
    e2.reset()
    with pytest.raises(ValueError, match="drift"):
        e2.step_replay(np.full((1, 4), 0.5, np.float32), 1.0, amd.ReplayTape(1, np.zeros((1, 4), np.int32), drift_uniforms=np.zeros((3, 1, 4))))
    e2.close()


def test_g1_nth_price_auction_on_gpu(amd, golden):
    import ctypes as C
    from adcraft_amd import _ffi
    L = _ffi.lib()
    for c in golden("g1_nth_price_auction.json")["cases"]:
        ob = np.array(c["other_bids"], dtype=np.float64)
        ob = ob.reshape(len(c["other_bids"]), -1) if len(c["other_bids"]) else np.zeros((0, 1))
        na, nb = ob.shape
        pl = np.zeros(max(na, 1), np.int32)
        co = np.zeros(max(na, 1), np.float64)
        imp = C.c_int32()
        _ffi.check(L.adc_nth_price_auction(0, c["bid"], ob.ctypes.data, na, nb, c["n"], c["num_winners"], C.byref(imp),
                                           pl.ctypes.data, co.ctypes.data))
        assert imp.value == c["impressions"], c["tag"]
        assert pl[:imp.value].tolist() == c["placements"], c["tag"]
        assert co[:imp.value].tolist() == c["costs"], c["tag"]


# ------------------------------------------------------------------ stream properties
def test_results_do_not_depend_on_sharding(amd):
    """envs [0,8) on one engine == envs [0,4) + [4,8) on two engines with env_id_base (multi-GPU invariant)"""
    N, K = 8, 96
    planes = H.implicit_params(N, K, seed=21)
    bids = np.random.default_rng(1).uniform(0.3, 1.0, (N, K)).astype(np.float32)
    whole = amd.StepEngine(N, K, seed=99)
    whole.set_all_params(planes)
    whole.reset()
    a = whole.step(bids, 1e9)
    parts = []
    for base in (0, 4):
        e = amd.StepEngine(4, K, seed=99, env_id_base=base)
        e.set_all_params(planes[:, base:base + 4])
        e.reset()
        parts.append(e.step(bids[base:base + 4], 1e9))
        e.close()
    for k in a:
        assert np.array_equal(a[k], np.concatenate([p[k] for p in parts])), k
    whole.close()


@pytest.mark.parametrize("budget", [1e9, 3.0])
def test_two_million_envs_of_few_keywords(amd, budget):
    """the env axis at 2^21 (index arithmetic, flagged lists and ticket counters far beyond the grid; narrow keyword tiles):
    the last 48 envs equal a 48-env engine placed at the same global ids, which in turn equals the oracle"""
    N, K, T = 1 << 21, 6, 48
    rng = np.random.default_rng(77)
    tail = H.implicit_params(T, K, seed=78, mean_volume=40)
    e = amd.StepEngine(N, K, seed=1234, max_days=3, auto_reset=True)
    row = H.implicit_params(1, K, seed=79, mean_volume=40)
    e.set_all_params(np.concatenate([np.broadcast_to(row, (8, N - T, K)), tail], axis=1))
    e.reset()
    small = amd.StepEngine(T, K, seed=1234, max_days=3, auto_reset=True, env_id_base=N - T)
    small.set_all_params(tail)
    small.reset()
    o = H.mirror_oracle(small, tail, max_days=3, auto_reset=True)
    bids_tail = np.round(rng.uniform(0.3, 1.0, (T, K)), 2).astype(np.float32)
    bids = np.full((N, K), 0.55, np.float32)
    bids[N - T:] = bids_tail
    for _ in range(4):                                   # crosses an episode end
        big, sm, ref = e.step(bids, budget, copy=False), small.step(bids_tail, budget), o.step(bids_tail, budget)
        H.assert_step_equal(sm, ref)
        for k in sm:
            assert np.array_equal(big[k][N - T:], sm[k]), k
        assert big["impressions"][: N - T].sum() > 0 and (big["days_passed"][: N - T] == big["days_passed"][0]).all()
    e.close()
    small.close()


def test_synthetic_actions_match_oracle(amd):
    N, K = 3, 130
    e = amd.StepEngine(N, K, seed=5)
    e.reset()
    o = H.mirror_oracle(e, H.implicit_params(N, K, 1))
    e.sample_actions(0.3, 1.0, 1e9)
    e.synchronize()
    import ctypes as C
    from adcraft_amd import _ffi
    p, nbytes = e.device_buffer(_ffi.BUF_BIDS)
    assert nbytes == N * K * 4
    # read the staging buffer back through a step: the bids the engine used are the oracle's
    e.set_all_params(o.params)
    e.step_device()
    got = e.fetch()
    ref = o.step(o.sample_bids(0.3, 1.0), 1e9)
    H.assert_step_equal(got, ref)
    e.close()


def test_step_before_reset_raises(amd):
    e = amd.StepEngine(1, 4)
    with pytest.raises(AssertionError):
        e.step(np.ones((1, 4), np.float32), 10.0)
    e.close()


# ------------------------------------------------------------------ BASELINE full size: size-independent properties
def _full_size_properties(amd, name, steps=1, seed=1730):
    """One BASELINE config at its full per-GPU size: domain properties that do not depend on the size (click <= impression,
    conversion <= click, no money without an event, the reward is the checksum of the per-keyword checksums), idempotence
    (same stream state + same actions -> the identical step), and a 4-env slice of the big launch against the CPU oracle on
    the same envs - for `steps` consecutive days (drift on where the config has it)."""
    from adcraft_amd import synthetic
    from oracle import capi as orc
    N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[name]
    planes = H.implicit_params(N, K, seed=seed, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
    e = amd.StepEngine(N, K, seed=seed, drift_enabled=drift)
    e.set_all_params(planes)
    e.reset()
    k0, t0 = e.get_rng_state()
    sub = slice(N // 3, N // 3 + 4)
    o = orc.OracleEngine(4, K, drift_on=drift)
    o.params[:] = planes[:, sub]
    o.key[:] = k0[sub]
    o.tick[:] = t0[sub]
    first = None
    vmax = mean_volume + 6 * (1 + mean_volume // 2)
    for s in range(steps):
        e.sample_actions(0.3, 1.0, 1e9)
        e.step_device()
        a = e.fetch()
        if first is None:
            first = a
        assert (a["buyside_clicks"] <= a["impressions"]).all()
        assert (a["sellside_conversions"] <= a["buyside_clicks"]).all()
        assert (a["cost"][a["buyside_clicks"] == 0] == 0).all() and (a["revenue"][a["sellside_conversions"] == 0] == 0).all()
        assert (a["impressions"] <= vmax * (1.0 + 0.2 * s)).all()
        assert a["impressions"].mean() > 0.12 * mean_volume * (1.0 - no_vol_prob)
        if no_vol_prob > 0:
            assert (a["impressions"][planes[0] == 0] <= 3).all()       # volume ~ N(0, U*0.5) clipped at 0: a rare auction or two
        cents = np.rint(a["revenue"].astype(np.float64) * 100) - np.rint(a["cost"].astype(np.float64) * 100)
        assert np.array_equal(np.rint(a["reward"] * 100), cents.sum(axis=1))          # checksum of checksums
        assert (a["days_passed"] == s + 1).all() and not a["terminated"].any()
        ref = o.step(o.sample_bids(0.3, 1.0), 1e9)
        H.assert_step_equal({k: v[sub] for k, v in a.items()}, ref)
    if drift:
        o.materialize_drift()
        got = e.get_all_params()
        assert np.array_equal(got[:, sub], o.params)
        assert not np.array_equal(got[0], planes[0])
    else:
        # idempotence: same stream state + same actions -> identical step
        e.set_rng_state(k0, t0)
        e.set_episode_state(np.zeros(N, np.int32), np.zeros(N))
        e.sample_actions(0.3, 1.0, 1e9)
        e.step_device()
        b = e.fetch()
        for k in first:
            assert np.array_equal(first[k], b[k]), k
    e.close()


def test_full_size_cfg2_properties(amd):
    """BASELINE configs[1]: 4096 envs x 256 keywords, dense stationary"""
    _full_size_properties(amd, "cfg2")


def test_full_size_cfg3_properties(amd):
    """BASELINE configs[2]: 16384 envs x 1024 keywords, sparse volume (half the keywords empty, mean volume 16, cvr 0.1)"""
    _full_size_properties(amd, "cfg3")


def test_full_size_cfg4_shard_properties(amd):
    """BASELINE configs[3]: the per-GPU shard (8192 x 1024) of 65536 x 1024 over 8 GPUs"""
    _full_size_properties(amd, "cfg4")


def test_full_size_cfg5_shard_properties_with_drift(amd):
    """BASELINE configs[4]: the per-GPU shard (2048 x 1024) of 16384 x 1024 with CTR/CVR/volume drift, three days"""
    _full_size_properties(amd, "cfg5", steps=3)


@pytest.mark.parametrize("shape,budget", [("cfg2", 1000.0), ("cfg2", 10.0), ((2048, 1024), 4000.0), ((2048, 1024), 40.0), ((4096, 512), 200.0)])
def test_full_size_binding_budgets(amd, shape, budget):
    """Every env's budget binding at BASELINE's full sizes (configs[1] and the K = 1024 / 512 shapes of the 8-GPU configs' shards),
    six days in a row, so that what the device tells the host between launches takes effect - the rest-of-day pair, envs parked
    at once: the domain's properties (nobody spends more than the budget; click <= impression, conversion <= click; the reward
    is the checksum of the per-keyword checksums) and a 4-env slice of the big launch against the CPU oracle on the same envs,
    bit for bit, every day."""
    from adcraft_amd import synthetic
    from oracle import capi as orc
    if isinstance(shape, str):
        N, K, mean_volume, cvr, no_vol_prob, _ = synthetic.CONFIGS[shape]
    else:
        (N, K), mean_volume, cvr, no_vol_prob = shape, 128, 0.8, 0.0
    planes = H.implicit_params(N, K, seed=1741, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
    e = amd.StepEngine(N, K, seed=1741, drift_enabled=True, max_days=1000, loss_threshold=1e12)
    e.set_all_params(planes)
    e.reset()
    e.direct_days(reset=True)
    k0, t0 = e.get_rng_state()
    sub = slice(N // 5, N // 5 + 4)
    o = orc.OracleEngine(4, K, drift_on=True, max_days=1000, loss_threshold=1e12)
    o.params[:] = planes[:, sub]
    o.key[:] = k0[sub]
    o.tick[:] = t0[sub]
    budget_c = int(np.rint(np.float64(np.float32(budget)) * 100))
    bound = 0
    for s in range(6):
        e.sample_actions(0.3, 1.0, budget)
        e.step_device()
        a = e.fetch()
        spend = np.rint(a["cost"].astype(np.float64) * 100).sum(axis=1)
        assert (spend <= budget_c).all()
        bound += int((spend >= budget_c - 150).sum())                  # (within a click or two of the budget)
        assert (a["buyside_clicks"] <= a["impressions"]).all() and (a["sellside_conversions"] <= a["buyside_clicks"]).all()
        assert (a["cost"][a["buyside_clicks"] == 0] == 0).all() and (a["revenue"][a["sellside_conversions"] == 0] == 0).all()
        cents = np.rint(a["revenue"].astype(np.float64) * 100) - np.rint(a["cost"].astype(np.float64) * 100)
        assert np.array_equal(np.rint(a["reward"] * 100), cents.sum(axis=1))          # checksum of checksums
        ref = o.step(o.sample_bids(0.3, 1.0), budget)
        H.assert_step_equal({k: v[sub] for k, v in a.items()}, ref)
    assert bound > 5 * N                                                # the budgets did bind
    if shape == "cfg2" and budget == 10.0:
        assert e.direct_days() > 2 * N                                  # ... and most days were parked at once
    o.materialize_drift()
    assert np.array_equal(e.get_all_params()[:, sub], o.params)
    e.close()


# ------------------------------------------------------------------ metrics on the device
def test_bid_curves_estimator_matches_reference(amd, golden):
    """k_ideal_profit's estimator on the reference's own samples == get_implicit_kw_bid_cpc_impressions (G5)"""
    from adcraft_amd import _ffi
    g = golden("g5_metrics.json")
    L = _ffi.lib()
    for c in g["bid_curves"]:
        s = np.array(c["samples_cents"], dtype=np.int32)
        ir = np.zeros(299)
        cpc = np.zeros(299)
        grid = np.array(g["bid_array"], dtype=np.float64)
        assert grid.size == 299
        _ffi.check(L.adc_bid_curves_from_samples(0, s.ctypes.data, s.size, grid.ctypes.data, 299, ir.ctypes.data, cpc.ctypes.data))
        assert ir.tolist() == c["impression_rates"]
        np.testing.assert_allclose(cpc, c["cpc"], rtol=1e-12)


def test_ideal_profit_matches_host_restatement(amd):
    from oracle import capi as orc, ref_numpy as rn
    N, K, n = 2, 6, 2048
    planes = H.implicit_params(N, K, seed=31)
    planes[0, 0, 1] = 0.0          # keywords whose profit is zero at every bid (no volume / no clicks / no margin): the kernel skips
    planes[4, 0, 2] = 0.0          # their sampling - the reference formula gives the same 0
    planes[5, 1, 3] = 0.0
    e = amd.StepEngine(N, K, seed=5)
    e.set_all_params(planes)
    e.reset()
    got = e.ideal_profit(n)
    keys, ticks = e.get_rng_state()
    bids = np.arange(0.01, 3.00, 0.01)
    L = orc.lib()
    for env in range(N):
        for k in range(K):
            cents = []
            for i in range(n // 4):
                w = orc.philox([i, 6, k, int(ticks[env])], [int(keys[env]) & 0xFFFFFFFF, int(keys[env]) >> 32])
                cents += [L.orc_competitor_cents_from_v(int(x) >> 8, float(planes[2, env, k]), float(planes[3, env, k])) for x in w]      # (stream revision 5)
            ir, cpc = rn.implicit_bid_cpc_impressions(np.array(cents, dtype=np.float64).reshape(1, -1) / 100.0, bids)
            kwp = [[float(planes[0, env, k]), 0.0], 0, 0, float(planes[4, env, k]), float(planes[5, env, k]), float(planes[6, env, k])]
            ref = rn.max_expected_bid_profits(kwp, cpc, ir)[0]
            assert got[env, k] == pytest.approx(ref, rel=1e-9, abs=1e-12)
    e.close()


def test_metric_accumulators(amd):
    N, K = 5, 70
    planes = H.implicit_params(N, K, seed=32)
    e = amd.StepEngine(N, K, seed=6, max_days=3, auto_reset=True)
    e.set_all_params(planes)
    e.reset()
    e.metrics_enable(True)
    e.metrics_reset()
    prof = np.zeros(K, dtype=np.int64)
    tot = 0
    rng = np.random.default_rng(0)
    for s in range(4):
        budget = 1e9 if s % 2 == 0 else 30.0          # exercise both passes
        out = e.step(rng.uniform(0.3, 1.0, (N, K)).astype(np.float32), budget)
        cents = np.rint(out["revenue"].astype(np.float64) * 100) - np.rint(out["cost"].astype(np.float64) * 100)
        prof += cents.sum(axis=0).astype(np.int64)
        tot += int(np.rint(out["reward"] * 100).sum())
    kp, sc = e.metrics_read()
    assert np.array_equal(kp, prof) and sc[0] == tot and sc[1] == 4 * N and sc[2] == N and sc[3] == 0
    e.close()


@pytest.mark.parametrize("variant", ["1", "2"])
def test_metric_accumulators_beyond_32_bits(amd, monkeypatch, variant):
    """the per-(env, keyword) profit sums are a 32-bit word + a 64-bit spill (common.inc metric_add): keywords whose revenue is
    millions of dollars a conversion overflow the word within a step or two - the totals stay exact (oracle cents), in both
    keyword-parallel kernels, through re-runs by the budget-exact kernels, and in every reader (columns, per-(env, keyword), median)"""
    monkeypatch.setenv("ADCRAFT_FAST_VARIANT", variant)
    N, K = 4, 300
    planes = H.implicit_params(N, K, seed=35, mean_volume=40)
    planes[6, :, ::3] = 4.0e6           # rev_mean in dollars: 4e8 cents per conversion
    planes[7, :, ::3] = 1.0e6
    planes[6, :, 1::3] = 9.0e6          # at the engine's money cap of 1e9 cents
    e = amd.StepEngine(N, K, seed=9)
    e.set_all_params(planes)
    e.reset()
    e.metrics_enable(True)
    e.metrics_reset()
    o = H.mirror_oracle(e, planes)
    total = np.zeros((N, K), np.int64)
    for budget in (1e9, 40.0, 1e9, 1e9, 40.0, 1e9):
        bids = o.sample_bids(0.3, 1.0)
        got, ref = e.step(bids, budget), o.step(bids, budget)
        assert np.array_equal(got["sellside_conversions"], ref["conversions"]) and np.array_equal(got["reward"], ref["reward"])
        total += ref["revenue_cents"] - ref["cost_cents"]
    assert np.abs(total).max() > 2 ** 33
    assert np.array_equal(np.rint(e.metrics_read_nk(ideal=False)[0] * 100).astype(np.int64), total)
    kp, sc = e.metrics_read()
    assert np.array_equal(kp, total.sum(axis=0)) and sc[0] == total.sum()
    e.close()


# ------------------------------------------------------------------ row-parallel exact pass: more shapes / edge budgets
@pytest.mark.parametrize("N,K,budget", [(3, 300, 120.0), (2, 1024, 400.0), (4, 257, 3.0), (3, 64, 0.0), (2, 64, -5.0),
                                        (2, 1500, 500.0)])
def test_exact_rows_shapes_and_edge_budgets(amd, N, K, budget):
    planes = H.implicit_params(N, K, seed=41 + K)
    _run_vs_oracle(amd, N, K, planes, steps=2, budget=budget, bid_lo=0.4, bid_hi=1.1)


@pytest.mark.parametrize("K", [300, 512, 513, 1024, 1500])
def test_row_kernel_beyond_256_keywords(amd, monkeypatch, K):
    """k_step_exact_rows beyond a lane per keyword: 256 lanes up to 512 keywords, 512 lanes beyond, several cells per lane and row -
    budgets that bind late, early, stop the campaign, are zero; drift; the rest of the day by the pair of kernels and by the one"""
    for split, budget, drift in ((1, 300.0, False), (0, 25.0, True), (1, 0.6, True), (1, 0.0, False)):
        monkeypatch.setenv("ADCRAFT_REST_SPLIT", str(split))
        planes = H.implicit_params(3, K, seed=70 + K, mean_volume=30)
        n = _run_vs_oracle(amd, 3, K, planes, steps=4, budget=budget, bid_lo=0.4, bid_hi=1.1, drift=drift, check_params=drift)
        assert n > 0


def test_exact_pass_sparse_and_large_cells(amd):
    planes = H.implicit_params(3, 200, seed=43, mean_volume=16, cvr=0.5, no_vol_prob=0.5)
    _run_vs_oracle(amd, 3, 200, planes, steps=3, budget=2.0)
    planes = H.implicit_params(2, 24, seed=44, mean_volume=2500)        # cells of ~100 auctions
    _run_vs_oracle(amd, 2, 24, planes, steps=2, budget=700.0)


@pytest.mark.parametrize("N,K,vol", [(6, 300, 128), (3, 2100, 12)])
def test_metric_accumulators_with_hinted_exact_pass(amd, N, K, vol):
    """metric sums stay exact when envs alternate between fast pass, re-run by the row kernel (K <= 2048) or by the serial
    walker (beyond), and hinted steps"""
    planes = H.implicit_params(N, K, seed=33, mean_volume=vol)
    e = amd.StepEngine(N, K, seed=8)
    e.set_all_params(planes)
    e.reset()
    e.metrics_enable(True)
    e.metrics_reset()
    prof = np.zeros(K, dtype=np.int64)
    rng = np.random.default_rng(1)
    for budget in (1e9, 50.0, 50.0, 1e9, 1e9, 20.0):
        out = e.step(rng.uniform(0.3, 1.0, (N, K)).astype(np.float32), budget)
        cents = np.rint(out["revenue"].astype(np.float64) * 100) - np.rint(out["cost"].astype(np.float64) * 100)
        prof += cents.sum(axis=0).astype(np.int64)
    kp, sc = e.metrics_read()
    assert np.array_equal(kp, prof) and sc[1] == 6 * N
    e.close()


@pytest.mark.parametrize("K", [2048, 4096])
def test_max_keyword_counts(amd, K):
    """K <= 2048: the row kernel with 73 B/keyword of dynamic LDS (146 KiB); beyond: the serial walker with 40 B/keyword
    (160 KiB at the K = 4096 limit)"""
    planes = H.implicit_params(1, K, seed=60 + K, mean_volume=12, cvr=0.5)
    _run_vs_oracle(amd, 1, K, planes, steps=2, budget=1e9)
    _run_vs_oracle(amd, 1, K, planes, steps=1, budget=30.0)


@pytest.mark.parametrize("K", [1500, 4096])
def test_explicit_model_at_large_keyword_counts(amd, K):
    """EXPLICIT: keyword-parallel pass, row walker (K = 1500: 81 KB of LDS) or, beyond its LDS, the serial walker"""
    planes = H.explicit_params(1, K, seed=70 + K)
    _run_vs_oracle(amd, 1, K, planes, steps=2, budget=1e9, model=1)
    _run_vs_oracle(amd, 1, K, planes, steps=1, budget=900.0, model=1)


def test_non_finite_inputs_do_not_break_the_step(amd):
    N, K = 2, 32
    planes = H.implicit_params(N, K, seed=61)
    planes[4, 0, 0] = np.nan          # ctr
    planes[7, 0, 1] = np.nan          # revenue std
    planes[3, 0, 2] = 0.0             # degenerate competitor scale
    bids = np.full((N, K), 0.8, np.float32)
    bids[0, 3] = np.nan
    bids[0, 4] = np.inf
    bids[1, 0] = -3.0
    e = amd.StepEngine(N, K, seed=2)
    e.set_all_params(planes)
    e.reset()
    o = H.mirror_oracle(e, planes)
    got, ref = e.step(bids, np.array([np.nan, 1e9], np.float32)), o.step(bids, np.array([np.nan, 1e9], np.float32))
    H.assert_step_equal(got, ref)
    assert got["buyside_clicks"][0].sum() == 0 and np.isfinite(got["reward"]).all()      # NaN budget pays nothing
    e.close()


def test_random_soak_short(amd):
    """a few seconds of tools/soak_parity.py: random shapes, laws, budgets, drift, autoreset - all bit-exact"""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "soak_parity.py"), "6", "7"], text=True, cwd=root, timeout=300)
    assert "soak ok" in out and "MISMATCH" not in out


def test_short_tape_is_an_error_not_a_fault(amd, golden):
    t = golden("g3_implicit_replay.json")["traces"][1]
    K = t["K"]
    e = amd.StepEngine(1, K, seed=1)
    e.reset()
    tp = t["tape"]
    for cut in ("bid", "click", "conv", "rev"):
        kw = dict(bid_cents=tp["bid"], click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"])
        name = {"bid": "bid_cents", "click": "click", "conv": "conv", "rev": "rev_cents"}[cut]
        kw[name] = kw[name][: len(kw[name]) // 2]
        tape = amd.ReplayTape(1, np.array(t["volumes"]).reshape(1, K), **kw)
        with pytest.raises(ValueError, match="tape exhausted"):
            e.step_replay(np.array(t["bids"], np.float32), t["budget"], tape)
    with pytest.raises(ValueError, match="volume out of range"):
        e.step_replay(np.array(t["bids"], np.float32), t["budget"],
                      amd.ReplayTape(1, np.full((1, K), -4), bid_cents=tp["bid"], click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"]))
    e.close()


# ------------------------------------------------------------------ the engine's own stream vs the reference's distributions (G9)
def test_step_outcome_distributions_match_the_reference(amd, golden):
    """G9: per-keyword means of every step output (and the reward) over thousands of days of the reference's own
    Python loop, vs the engine's Philox stream on the same keywords and bids - including binding budgets, where
    the order-dependent budget walk shapes the distribution.  z-test at 5 sigma on the difference of means."""
    for sc in golden("g9_step_statistics.json")["scenarios"]:
        K, N = sc["K"], 8192
        planes = np.zeros((8, N, K), np.float32)
        for k, p in enumerate(sc["keyword_params"]):
            planes[:, :, k] = np.array([p["vol_mean"], p["vol_std"], p["loc"], p["scale"], p["bctr"], p["sctr"], p["rev_mean"],
                                        p["rev_std"]], np.float32)[:, None]
        e = amd.StepEngine(N, K, seed=2024, max_days=1 << 30, loss_threshold=1e12)
        e.set_all_params(planes)
        e.reset()
        bids = np.tile(np.array(sc["bids"], np.float32), (N, 1))
        acc = {k: [] for k in ("impressions", "buyside_clicks", "cost", "sellside_conversions", "revenue", "reward")}
        for _ in range(4):
            out = e.step(bids, sc["budget"])
            for k in acc:
                acc[k].append(out[k].astype(np.float64))
        e.close()
        n_eng, n_ref = 4 * N, sc["steps"]
        for name in ("impressions", "buyside_clicks", "cost", "sellside_conversions", "revenue"):
            x = np.concatenate(acc[name], axis=0)
            se = np.sqrt(np.array(sc["var"][name]) / n_ref + x.var(axis=0, ddof=1) / n_eng)
            z = (x.mean(axis=0) - np.array(sc["mean"][name])) / np.maximum(se, 1e-9)
            assert np.abs(z).max() < 5.0, (sc["name"], name, z.tolist())
            # ... and the per-keyword VARIANCES (a stream whose neighbouring counters were correlated would keep the means and
            # change the spread of a day's sums): log-ratio of the two sample variances, its standard error from the
            # engine's own fourth moment (kurtosis k: var(s^2) ~ s^4 (k - 1) / n)
            v_ref, v_eng = np.array(sc["var"][name]), x.var(axis=0, ddof=1)
            ok = (v_ref > 0) & (v_eng > 0)
            d = x - x.mean(axis=0)
            kurt = np.where(ok, (d**4).mean(axis=0) / np.maximum(d.var(axis=0), 1e-30) ** 2, 3.0)
            se_log = np.sqrt(np.maximum(kurt - 1.0, 0.5) * (1.0 / n_eng + 1.0 / n_ref))
            zv = np.where(ok, np.log(np.maximum(v_eng, 1e-30) / np.maximum(v_ref, 1e-30)) / se_log, 0.0)
            assert np.abs(zv).max() < 5.0, (sc["name"], name, "variance", zv.tolist())
        r = np.concatenate(acc["reward"])
        z = (r.mean() - sc["reward_mean"]) / np.sqrt(sc["reward_var"] / n_ref + r.var(ddof=1) / n_eng)
        assert abs(z) < 5.0, (sc["name"], "reward", z)
        if sc["budget"] < 1e8:     # the budget really binds, and is never overspent
            spend = np.concatenate(acc["cost"], axis=0).sum(axis=1)
            assert spend.max() <= sc["budget"] + 1e-3 and (spend > sc["budget"] - 1.5).mean() > 0.2
