"""GPU parity of the device-resident callers of the step: NaiveZeroMarginStrategy, the oracle
bidder and the per-step ideal profit, against the reference's recorded runs (G10, G5) and the oracle restatement."""
import numpy as np
import pytest

from oracle import capi as orc
from oracle import ref_numpy as rn
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import adcraft_amd.engine as eng
    from adcraft_amd import _ffi
    assert _ffi.device_count() >= 1, "no HIP device visible: the engine has no CPU path"
    return eng


def _uniform53(a, b):
    return float(((int(a) >> 5) << 26) | (int(b) >> 6)) * 2.0 ** -53


def test_zero_margin_agent_replays_the_reference_runs(amd, golden):
    """G10 through the C ABI: observations and the uniforms the reference drew in, the reference's action out -
    bids equal to the cent (what the env makes of them, gymnasium_kw_env.py:215), budget and caches bit for bit."""
    from adcraft_amd.baselines.interpolated_expectations import NaiveZeroMarginStrategy
    for c in golden("g10_zero_margin_agent.json")["cases"]:
        K = c["K"]
        agent = NaiveZeroMarginStrategy(K, default_expected_revenue_per_conversion=c["default_rpc"], seed=c["agent_seed"])
        action = {"budget": 0.0, "keyword_bids": 0.01 + np.zeros(K)}
        for t, s in enumerate(c["steps"]):
            obs = dict(buyside_clicks=np.array(s["obs_clicks"]), sellside_conversions=np.array(s["obs_conversions"]),
                       revenue=np.array(s["obs_revenue"], np.float32))
            agent.update_all_caches(action, obs)
            u = np.array(s["uniforms"])
            action = agent.sample_action(replay_uniforms=np.where(np.isnan(u), 0.5, u))
            want_cents = np.maximum(np.rint(np.array(s["bids"]) * 100.0), 1.0)
            assert np.array_equal(np.rint(action["keyword_bids"] * 100.0), want_cents), (t, action["keyword_bids"], s["bids"])
            assert action["budget"] == s["budget"]
            caches = agent.caches
            assert [cc["ave_rpc"] for cc in caches] == s["ave_rpc"]
            assert [cc["num_rpc_obs"] for cc in caches] == s["num_rpc_obs"]
            assert [cc["num_sctr_obs"] for cc in caches] == s["num_sctr_obs"]
            for cc, want, n in zip(caches, s["ave_sctr"], s["num_sctr_obs"]):
                assert n == 0 or cc["ave_sctr"] == want
            assert np.array_equal(agent.max_bids, np.array(s["max_bids"]))
        agent.close()


def test_zero_margin_agent_closed_loop_matches_the_restatement(amd):
    """the agent on its own Philox stream, fed by the engine's device-resident observations, vs the numpy restatement
    fed the fetched observations and the same uniforms (Philox call (0, AGENT, keyword, agent tick))"""
    N, K, steps = 3, 70, 12
    planes = H.implicit_params(N, K, seed=91, mean_volume=6, cvr=0.5)
    e = amd.StepEngine(N, K, seed=17, max_days=steps)
    e.set_all_params(planes)
    e.reset()
    seeds = np.array([5, 6, 7], np.uint64)
    e.agent_init(1.0, seeds)
    ref = rn.ZeroMarginAgent(N, K, 1.0)
    # the agent's key: splitmix64(seed ^ const) - read back through the uniforms it must have used: recompute here
    def splitmix64(x):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    keys = [splitmix64(int(s) ^ 0xA6E27B1D5C3F9E01) for s in seeds]
    obs = dict(buyside_clicks=np.zeros((N, K)), sellside_conversions=np.zeros((N, K)), revenue=np.zeros((N, K), np.float32))
    for t in range(steps):
        e.agent_step(100000.0)
        bids, budget = e.get_actions()
        u = np.zeros((N, K))
        for n in range(N):
            for k in range(K):
                w = orc.philox([0, 9, k, t], [keys[n] & 0xFFFFFFFF, keys[n] >> 32])
                u[n, k] = _uniform53(w[0], w[1])
        ref.update(obs["buyside_clicks"], obs["sellside_conversions"], obs["revenue"])
        rb, _ = ref.act(u)
        assert np.array_equal(np.rint(bids.astype(np.float64) * 100.0), np.maximum(np.rint(rb * 100.0), 1.0)), t
        assert np.all(budget == 100000.0)
        st = e.agent_state()
        assert np.array_equal(st["ave_rpc"], ref.ave_rpc) and np.array_equal(st["num_rpc_obs"], ref.num_rpc_obs)
        assert np.array_equal(st["num_sctr_obs"], ref.num_sctr_obs) and np.array_equal(st["max_bids"], ref.max_bids)
        assert np.array_equal(st["ave_sctr"][ref.num_sctr_obs > 0], ref.ave_sctr[ref.num_sctr_obs > 0])
        e.step_device()
        obs = e.fetch()
    e.close()


def test_splitmix_constant_matches_the_kernel():
    """(documents the agent key derivation used above; the value is checked by the closed-loop test itself)"""
    assert 0xA6E27B1D5C3F9E01 < 2 ** 64


def test_ideal_step_is_the_reference_formula_on_the_cached_curves(amd, golden):
    """per-step ideal under drift: get_max_expected_bid_profits (experiment_metrics.py:40-61) on the curves cached at
    reset and the CURRENT drifted parameters; value and argmax vs the numpy restatement pinned by G5"""
    g5 = golden("g5_metrics.json")
    for m in g5["max_expected"]:          # the restatement itself against the reference's recorded answers
        r = rn.max_expected_bid_profits(m["kw_params"], np.array(m["cpc"]), np.array(m["ir"]))
        assert r[0] == m["max_profit"] and r[2] == m["argmax"]
    N, K = 4, 33
    planes = H.implicit_params(N, K, seed=92, mean_volume=40, cvr=0.6)
    e = amd.StepEngine(N, K, seed=23, drift_enabled=True, drift=(0.2, 0.2, 0.2), max_days=50)
    e.set_all_params(planes)
    e.reset()
    e.bid_curves_build(2048)
    ir, cpc = e.bid_curves_fetch()
    assert np.array_equal(e.ideal_step()[0], e.ideal_profit(2048))     # same samples, same estimator, no drift yet
    e.sample_actions(0.3, 1.0, 1e9)
    for _ in range(3):
        e.step_device()
        ideal, best = e.ideal_step()
        p = e.get_all_params()            # (ideal_step has applied the pending drift, as the reference's update_keywords did)
        for n in range(N):
            for k in range(K):
                kwp = [[float(p[0][n, k]), 0.0], 0.0, 0.0, float(p[4][n, k]), float(p[5][n, k]), float(p[6][n, k]), 0.0]
                mx, _, am = rn.max_expected_bid_profits(kwp, cpc[n, k], ir[n, k])
                assert ideal[n, k] == mx and (best[n, k] == am or mx == 0.0), (n, k)
    assert not np.array_equal(p[0], planes[0])      # the volume means did drift
    e.close()


@pytest.mark.parametrize("case", ["notebook", "few_samples", "coarse_grid", "fine_grid", "extremes"])
def test_contender_lists_give_the_full_scans_ideal_bit_for_bit(amd, monkeypatch, case):
    """the per-step ideal evaluated on the contender grid points only (k_curve_contenders + k_ideal_from_contenders) against
    the same engine evaluating the WHOLE grid (ADCRAFT_IDEAL_FULL_SCAN=1, k_ideal_from_curves - itself pinned to the
    reference's formula above): value and first argmax identical for every keyword on every day of a strongly drifting episode"""
    N, K, days = 24, 160, 12
    n_samples, grid = 2048, None
    planes = H.implicit_params(N, K, seed=96, mean_volume=30, cvr=0.6)
    if case == "few_samples":
        n_samples = 24                                    # long runs of identical lines, many exact ties
    elif case == "coarse_grid":
        grid = np.arange(0.05, 2.0, 0.05)
    elif case == "fine_grid":
        grid = np.arange(0.002, 2.5, 0.002)               # 1249 points: the stack does not fit -> whole grid, still identical
    elif case == "extremes":
        planes[5, 0] = 0.0                                # no conversions: every profit <= 0 -> index 0
        planes[5, 1] = 1.0
        planes[6, 2] = 2.0e6                              # margin beyond kMarginMax -> whole grid
        planes[6, 3] = 1e-6
        planes[4, 4] = 0.0                                # no clicks
        planes[0, 5] = 0.0                                # no volume
        planes[6, 6] = np.nan
        planes[3, 7] = 1e-4                               # competitor bids all equal: two distinct lines
    runs = []
    for full in ("1", "0"):
        monkeypatch.setenv("ADCRAFT_IDEAL_FULL_SCAN", full)
        e = amd.StepEngine(N, K, seed=37, drift_enabled=True, drift=(0.3, 0.3, 0.5), max_days=1 << 20, loss_threshold=1e12)
        e.set_all_params(planes)
        e.reset()
        e.bid_curves_build(n_samples, grid)
        e.sample_actions(0.3, 1.0, 1e9)
        out = []
        for _ in range(days):
            ideal, best = e.ideal_step()
            out.append((ideal.copy(), best.copy()))
            e.step_device()
        runs.append(out)
        e.close()
    for (i_full, b_full), (i_fast, b_fast) in zip(*runs):
        assert np.array_equal(i_full, i_fast, equal_nan=True)
        assert np.array_equal(b_full, b_fast)
    assert len({tuple(b.ravel()) for _, b in runs[0]}) > 1 or case == "extremes"      # the argmax really moved with the drift


@pytest.mark.parametrize("n_samples,grid", [(2048, None), (40, None), (512, np.arange(0.01, 3.01, 0.01))])
def test_contender_lists_hold_the_argmax_at_every_margin(amd, n_samples, grid):
    """k_curve_contenders on its own (the envelope by wave-parallel elimination, the intervals, the scans that make their ends monotone):
    for every keyword and a sweep of margins m, the FIRST argmax over the whole grid of ir_b (m - cpc_b) - numpy, float64, the reference's
    expression (experiment_metrics.py:51-61) on the fetched curves - is a listed grid point whose margin interval holds m, the lists are
    in grid order, and the points whose interval holds a margin are one contiguous run of the list"""
    N, K = 6, 96
    planes = H.implicit_params(N, K, seed=5, mean_volume=30, cvr=0.6)
    planes[3, 1] *= np.linspace(0.1, 4.0, K).astype(np.float32)          # a spread of competitor laws
    e = amd.StepEngine(N, K, seed=3)
    e.set_all_params(planes)
    e.reset()
    e.bid_curves_build(n_samples, grid)
    ir, cpc = e.bid_curves_fetch()
    count, idx, iv = e.bid_curves_contenders()
    e.close()
    assert (count != 65535).mean() > 0.9 and count[count != 65535].max() > 2      # (65535: more distinct contenders than a list holds - the whole grid)
    margins = np.concatenate([np.linspace(0.0, 3.5, 701), np.exp(np.linspace(np.log(1e-4), np.log(50.0), 300))])
    checked = 0
    for n in range(N):
        for k in range(K):
            c = int(count[n, k])
            if c == 65535:
                continue
            ids, lo, hi = idx[n, k, :c], iv[n, k, :c, 0].astype(np.float64), iv[n, k, :c, 1].astype(np.float64)
            assert (np.diff(ids) > 0).all()
            assert (lo[1:] >= lo[:-1]).all() and (hi[1:] >= hi[:-1]).all()        # monotone ends (infinite ones included): the holders of a margin are contiguous
            profit = ir[n, k][None, :] * (margins[:, None] - cpc[n, k][None, :])      # [margins, grid]
            profit = np.where(profit > 0.0, profit, 0.0)
            best = profit.argmax(axis=1)                                              # np.argmax: the first maximum
            pos = profit.max(axis=1) > 0.0
            for m, b in zip(margins[pos], best[pos]):
                j = np.searchsorted(ids, b)
                assert j < c and ids[j] == b, (n, k, m, b)
                assert lo[j] <= m <= hi[j], (n, k, m, b, lo[j], hi[j])
                checked += 1
    assert checked > 10000


def test_baseline_episode_metrics_match_a_step_by_step_run(amd):
    """run_baseline_episode (everything device-resident) vs the same loop done the notebooks' way: fetch every
    observation, stack kw_profits / ideal_profits, compute_AKNCP / compute_NCP (restatements pinned by G5)"""
    from adcraft_amd.closed_loop import run_baseline_episode
    N, K, steps = 5, 40, 15
    planes = H.implicit_params(N, K, seed=93, mean_volume=24, cvr=0.7)
    for policy in ("zero_margin", "oracle"):
        runs = []
        for fused in (True, False):
            e = amd.StepEngine(N, K, seed=29, drift_enabled=True, drift=(0.03, 0.03, 0.03), max_days=steps)
            e.set_all_params(planes)
            e.reset()
            if fused:
                r = run_baseline_episode(e, policy, steps=steps, budget=100000.0, default_rpc=1.0, agent_seeds=np.arange(N, dtype=np.uint64))
                runs.append((r["AKNCP"], r["NCP"]))
                # the same two metrics reduced on the device (per-env median over the keywords in LDS): the median is the
                # host's bit for bit (same operations, a sort instead of a partition), NCP to the last bits of the sum order
                a_dev, n_dev = e.metrics_akncp_ncp(steps)
                assert np.array_equal(a_dev, r["AKNCP"]), policy
                assert np.allclose(n_dev, r["NCP"], rtol=1e-12, atol=0), policy
            else:
                e.bid_curves_build(2048)
                if policy == "zero_margin":
                    e.agent_init(1.0, np.arange(N, dtype=np.uint64))
                prof, ideal = [], []
                for _ in range(steps):
                    if policy == "zero_margin":
                        e.agent_step(100000.0)
                        ideal.append(e.ideal_step()[0])
                    else:
                        ideal.append(e.ideal_step()[0])
                        e.policy_oracle(100000.0)
                    e.step_device()
                    o = e.fetch()
                    prof.append(o["revenue"].astype(np.float64) - o["cost"].astype(np.float64))
                prof, ideal = np.array(prof), np.array(ideal)          # [T, N, K]
                runs.append((np.array([rn.compute_AKNCP(prof[:, n], ideal[:, n]) for n in range(N)]),
                             np.array([rn.compute_NCP(prof[:, n], ideal[:, n]) for n in range(N)])))
            e.close()
        assert np.allclose(runs[0][0], runs[1][0], rtol=1e-5, atol=1e-7), policy     # float32 dollars in the fetched obs
        assert np.allclose(runs[0][1], runs[1][1], rtol=1e-5, atol=1e-7), policy
    # the oracle bidder earns about the ideal (NCP near 1), the ramping agent less in 15 days
    assert 0.6 < np.median(runs[0][1]) < 1.4


def test_policy_calls_between_chained_steps_run_group_by_group(amd):
    """agent_step / ideal_step(fetch=False) / policy_oracle / sample_actions between two device-resident steps do not end a chain: while
    the steps run as env groups they are launched per group on the groups' streams (no fork, no join).  Same trajectory, agent state,
    ideal sums and drifted parameters as the one-stream schedule; a fetch in the middle (which joins) changes nothing either"""
    N, K = 9, 64
    planes = H.implicit_params(N, K, seed=96, mean_volume=30, cvr=0.6)
    out = []
    for groups in (1, 3):
        e = amd.StepEngine(N, K, seed=45, drift_enabled=True, drift=(0.05, 0.05, 0.05), max_days=1 << 20, loss_threshold=1e12)
        e.set_env_groups(groups)
        e.set_all_params(planes)
        e.reset()
        e.metrics_enable(True)
        e.bid_curves_build(2048)
        e.agent_init(1.0, np.arange(N, dtype=np.uint64))
        for day in range(12):
            if day < 6:
                e.agent_step(30.0)
                e.ideal_step(fetch=False)
            elif day < 9:
                e.ideal_step(fetch=False)
                e.policy_oracle(30.0)
            else:
                e.sample_actions(0.3, 1.0, 30.0)
            e.step_device()
            if day == 4:
                mid = e.fetch()
        assert e.env_groups() == groups
        o = e.fetch()
        st = e.agent_state()
        profit, ideal, ideal_pos = e.metrics_read_nk()
        out.append((o, mid, st, profit, ideal, ideal_pos, e.get_all_params(), e.get_actions()))
        e.close()
    a, b = out
    for i in (0, 1, 2):
        for key in a[i]:
            assert np.array_equal(a[i][key], b[i][key], equal_nan=True), (i, key)
    for i in (3, 4, 5, 6):
        assert np.array_equal(a[i], b[i]), i
    assert np.array_equal(a[7][0], b[7][0]) and np.array_equal(a[7][1], b[7][1])


@pytest.mark.parametrize("policy", ["zero_margin", "oracle", "fixed"])
def test_run_days_in_env_groups_equals_one_group(amd, policy):
    """adc_engine_run_days with the step in env groups on their own streams: every day's policy kernels write the bids on the engine's
    stream, so every day's groups have to wait for it - same trajectory as the one-stream schedule (which other tests pin to the oracle),
    with and without the per-step ideal between the agent and the step"""
    N, K, days = 9, 64, 14
    planes = H.implicit_params(N, K, seed=95, mean_volume=30, cvr=0.6)
    for curves in ((False, True) if policy == "zero_margin" else (policy == "oracle",)):
        out = []
        for groups in (1, 3):
            e = amd.StepEngine(N, K, seed=41, drift_enabled=True, drift=(0.05, 0.05, 0.05), max_days=1 << 20, loss_threshold=1e12)
            e.set_env_groups(groups)
            e.set_all_params(planes)
            e.reset()
            e.metrics_enable(True)
            if curves:
                e.bid_curves_build(2048)
            if policy == "zero_margin":
                e.agent_init(1.0, np.arange(N, dtype=np.uint64))
            else:
                e.sample_actions(0.3, 1.0, 40.0)
            e.run_days(policy, days, budget=40.0, graph=False)
            assert e.env_groups() == groups
            o = e.fetch()
            kp, sc = e.metrics_read()
            out.append((o, kp, sc, e.get_all_params(), e.get_actions()))
            e.close()
        (a, akp, asc, ap, aa), (b, bkp, bsc, bp, ba) = out
        for key in a:
            assert np.array_equal(a[key], b[key]), (policy, curves, key)
        assert np.array_equal(akp, bkp) and np.array_equal(asc, bsc) and np.array_equal(ap, bp)
        assert np.array_equal(aa[0], ba[0]) and np.array_equal(aa[1], ba[1])


@pytest.mark.parametrize("K", [1, 2, 33, 64, 1000, 4096])
def test_device_median_matches_numpy_for_odd_even_and_padded_keyword_counts(amd, K):
    from adcraft_amd.closed_loop import run_baseline_episode
    N, steps = 3, 3
    planes = H.implicit_params(N, K, seed=94 + K, mean_volume=6, cvr=0.6, no_vol_prob=0.3)
    e = amd.StepEngine(N, K, seed=31, max_days=steps)
    e.set_all_params(planes)
    e.reset()
    r = run_baseline_episode(e, "oracle", steps=steps, budget=100000.0, n_samples=256)
    a_dev, n_dev = e.metrics_akncp_ncp(steps)
    assert np.array_equal(a_dev, r["AKNCP"]) and np.allclose(n_dev, r["NCP"], rtol=1e-12, atol=0)
    r2 = run_baseline_episode(e, "oracle", steps=steps, budget=100000.0, n_samples=256, per_keyword_sums=False)      # a second episode: device-only
    assert set(r2) == {"AKNCP", "NCP"} and r2["AKNCP"].shape == (N,)
    e.close()


def test_device_median_propagates_nan_like_numpy(amd):
    """np.median of a ratio vector that holds a NaN is NaN (closed_loop.py's host path); k_akncp_ncp sorts a NaN as +inf and must
    still report NaN for that env, so that the device-side reduction and the host's agree.  Here: days stepped without any ideal
    profit accumulated (denominators 0) and a keyword without auctions in env 1 (profit 0): 0 / 0 among +-inf."""
    N, K, steps = 3, 33, 3
    planes = H.implicit_params(N, K, seed=77, mean_volume=6, cvr=0.6, no_vol_prob=0.0)
    planes[0, 1, 5] = 0.0           # vol_mean
    planes[1, 1, 5] = 0.0           # vol_std
    e = amd.StepEngine(N, K, seed=31, max_days=1000)
    e.set_all_params(planes)
    e.reset()
    e.bid_curves_build(256)
    e.metrics_enable(True)
    e.ideal_step()
    e.metrics_reset()               # the sums are zero again
    e.sample_actions(0.3, 1.0, 100000.0)
    for _ in range(steps):
        e.step_device()
    profit, ideal, ideal_pos = e.metrics_read_nk()
    assert profit[1, 5] == 0.0 and not ideal_pos.any()
    with np.errstate(divide="ignore", invalid="ignore"):
        host = np.median((profit / steps) / (ideal_pos / steps), axis=1)
    assert np.isnan(host[1])
    a_dev, _ = e.metrics_akncp_ncp(steps)
    assert np.array_equal(a_dev, host, equal_nan=True)
    e.close()


@pytest.mark.parametrize("model", [0, 1])
def test_run_days_graph_replay_equals_single_steps(amd, model):
    """adc_engine_run_days, plain and replaying pairs of days from a captured hipGraph: same trajectory as one call per
    day, for odd and even day counts, repeated calls (parity re-alignment) and both models"""
    N, K = 6, 48
    planes = H.implicit_params(N, K, seed=95, mean_volume=20, cvr=0.6) if model == 0 else H.explicit_params(N, K, seed=95)
    outs = []
    for graph in (True, False, None):
        e = amd.StepEngine(N, K, model=model, seed=31, drift_enabled=True, drift=(0.05, 0.05, 0.05), max_days=1 << 20,
                           loss_threshold=1e12)
        e.set_all_params(planes)
        e.reset()
        e.sample_actions(0.3, 1.0, 35.0)              # a budget that binds on some days
        e.metrics_enable(True)
        for days in (1, 7, 4, 5, 2):
            if graph is not None:
                e.run_days("fixed", days, graph=graph)
            else:
                for _ in range(days):
                    e.step_device()
        o = e.fetch()
        outs.append((o, e.get_all_params(), e.metrics_read_nk(ideal=False)[0], e.get_rng_state()))
        e.close()
    for a in outs[:2]:
        b = outs[2]
        for k in a[0]:
            assert np.array_equal(a[0][k], b[0][k]), k
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.array_equal(a[3][0], b[3][0]) and np.array_equal(a[3][1], b[3][1])


@pytest.mark.parametrize("budget", [2.0, 60.0])
def test_run_days_graph_replay_with_the_rest_of_day_kernels(amd, monkeypatch, budget):
    """The same with every budget binding and the pair of rest-of-day kernels forced (ADCRAFT_REST_SPLIT=1; the default
    takes it from 1024 envs on): the graphs are captured before and after the device has told the host that budgets bind and
    that envs are parked at once (budget 2), and must be re-captured when it has; memory those paths need is allocated between
    captures, never inside one.  Same trajectory as one call per day."""
    monkeypatch.setenv("ADCRAFT_REST_SPLIT", "1")
    monkeypatch.setenv("ADCRAFT_CLICK_WALK", "0")
    N, K = 6, 200
    planes = H.implicit_params(N, K, seed=97, mean_volume=40, cvr=0.6)
    outs = []
    for graph in (True, False, None):
        e = amd.StepEngine(N, K, seed=33, drift_enabled=True, drift=(0.05, 0.05, 0.05), max_days=1 << 20, loss_threshold=1e12)
        e.set_all_params(planes)
        e.reset()
        e.direct_days(reset=True)
        e.sample_actions(0.4, 1.2, budget)
        e.metrics_enable(True)
        for days in (5, 6, 7, 4, 9):
            if graph is not None:
                e.run_days("fixed", days, graph=graph)
                e.synchronize()                       # (the host reads the device's flags between calls)
            else:
                for _ in range(days):
                    e.step_device()
                e.synchronize()
        o = e.fetch()
        outs.append((o, e.get_all_params(), e.metrics_read_nk(ideal=False)[0], e.get_rng_state(), e.direct_days()))
        e.close()
    for a in outs[:2]:
        b = outs[2]
        for k in a[0]:
            assert np.array_equal(a[0][k], b[0][k]), k
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.array_equal(a[3][0], b[3][0]) and np.array_equal(a[3][1], b[3][1])
    if budget == 2.0:
        assert outs[0][4] > 0 and outs[2][4] > 0      # envs were parked at once, under graph replay too


def test_day_graph_is_recaptured_when_settings_change(amd):
    """The kernel nodes of a captured day hold the engine's settings by value.  Changing any of them between two
    run_days(graph=True) calls - metrics on/off, episode limits, drift, flat observations, the bid-curve grid (which
    re-allocates device memory the old graph would still point at) - must give what the un-captured loop gives."""
    N, K = 5, 40
    planes = H.implicit_params(N, K, seed=96, mean_volume=24, cvr=0.5)
    outs = []
    for graph in (True, False):
        e = amd.StepEngine(N, K, seed=32, max_days=1 << 20, loss_threshold=1e12, auto_reset=True)
        e.set_all_params(planes)
        e.reset()
        e.bid_curves_build(512, np.arange(0.01, 3.00, 0.01))
        e.metrics_enable(False)
        e.run_days("oracle", 6, budget=1e6, graph=graph)
        e.metrics_enable(True)                              # was off when the graph was captured
        e.metrics_reset()
        e.run_days("oracle", 6, budget=1e6, graph=graph)
        e.bid_curves_build(256, np.arange(0.05, 2.00, 0.05))   # another grid size: the curve buffers are re-allocated
        e.run_days("oracle", 6, budget=1e6, graph=graph)
        e.set_limits(4, 1e12)                               # episodes now end (auto-reset) every 4 days
        e.set_drift(True, (0.05, 0.05, 0.05))
        e.run_days("oracle", 9, budget=1e6, graph=graph)
        o = e.fetch()
        prof, ideal, ideal_pos = e.metrics_read_nk(ideal=True)
        outs.append((o, e.get_all_params(), prof, ideal, e.metrics_read()[1]))
        e.close()
    a, b = outs
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert a[4][1] == 21 * N and a[4][2] > 0                # metric mode counted the 21 days after it was switched on; episodes ended


def test_heatmap_cells_match_the_reference_end_to_end(amd, golden):
    """G11: cells of the paper's heat-map experiment run by the reference itself (its env, its agent, its metrics, the
    notebook's loop).  The device-resident loop on the same keyword sets (reset(seed) generation is bit-exact) must give
    the same mean AKNCP / NCP / profit per cell within the run-to-run noise: z-test on the difference of cell means."""
    from adcraft_amd import gymnasium_kw_utils as utils
    from adcraft_amd.closed_loop import run_baseline_episode
    for cell in golden("g11_heatmap_cells.json")["cells"]:
        K, days = cell["K"], cell["days"]
        cfg = utils.experiment_keyword_config(cell["mean_volume"], cell["cvr"])
        env_seeds = sorted(int(s) for s in cell["keyword_params"])
        reps = 16                                        # engine runs per env seed (the reference has 4)
        N = len(env_seeds) * reps
        planes = np.zeros((8, N, K), np.float32)
        for i, es in enumerate(env_seeds):
            rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(es)))
            raw = utils.sample_implicit_keyword_params(K, rng, cfg)
            printed = utils.printed_params(raw, implicit=True)
            got = [[[float(p[0][0]), float(p[0][1])]] + [float(x) for x in p[1:]] for p in printed]
            assert got == cell["keyword_params"][str(es)]              # the very keyword set the reference ran on
            planes[:, i * reps:(i + 1) * reps] = utils.implicit_params_to_planes(raw)[:, None, :]
        e = amd.StepEngine(N, K, seed=77, max_days=days, loss_threshold=10000.0, drift_enabled=bool(cell.get("drift")),
                           drift=(0.03, 0.03, 0.03))
        e.set_all_params(planes)
        e.reset(seeds=np.arange(N, dtype=np.uint64) + 1000)
        r = run_baseline_episode(e, "zero_margin", steps=days, budget=100000.0, default_rpc=1.0,
                                 agent_seeds=np.arange(N, dtype=np.uint64))
        # the oracle bidder on the same keyword sets (run_oracle_agent, 300-point grid)
        e.set_all_params(planes)                          # (a drift cell: every reference run starts from the fresh keyword set)
        e.reset(seeds=np.arange(N, dtype=np.uint64) + 9000)
        ro = run_baseline_episode(e, "oracle", steps=days, budget=100000.0, bid_grid=np.arange(0.01, 3.01, 0.01))
        e.close()
        for ref, res, who in ((cell["runs"], r, "zero_margin"), (cell["oracle_runs"], ro, "oracle")):
            assert all(x["days"] == days for x in ref)
            for name, mine in (("AKNCP", res["AKNCP"]), ("NCP", res["NCP"]), ("total_profit", res["kw_profit_sum"].sum(axis=1))):
                theirs = np.array([x[name] for x in ref])
                # under the hypothesis tested both samples have the same per-run spread; the engine's many runs estimate
                # it far better than the reference's 8 or 16 (whose sample variance alone makes a fragile z)
                se = mine.std(ddof=1) * np.sqrt(1.0 / theirs.size + 1.0 / mine.size)
                z = (mine.mean() - theirs.mean()) / se
                assert abs(z) < 4.0, (cell["mean_volume"], cell["cvr"], cell.get("drift"), who, name, float(mine.mean()), float(theirs.mean()), float(z))
        ref = cell["runs"]
        # the ideal profit is a property of the keyword set alone (same env seed -> same expected optimum, up to the
        # 2048-sample estimator's noise)
        for i, es in enumerate(env_seeds):
            theirs = np.mean([x["total_ideal"] for x in ref if x["env_seed"] == es])
            mine = r["ideal_sum"][i * reps:(i + 1) * reps].sum(axis=1).mean()
            # (a drift cell: the ideal follows each run's own random walk of the parameters, so the 4-run mean is noisier)
            assert abs(mine / theirs - 1.0) < (0.15 if cell.get("drift") else 0.05), (es, mine, theirs)
