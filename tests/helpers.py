"""Shared helpers for the parity tests: synthetic keyword sets (the BASELINE configs' laws),
and oracle <-> engine state mirroring."""
import numpy as np

from oracle import capi as orc


def _pl(rng, lo, mid, hi, shape):
    """piecewise-linear quantile interpolation of a single (min, median, max) bucket
    (adcraft/pull_quantiles_data/quantiles_to_keywords.py:24-27)"""
    return np.interp(rng.random(shape), [0.0, 0.5, 1.0], [lo, mid, hi])


def implicit_params(N, K, seed, mean_volume=128, cvr=0.8, no_vol_prob=0.0):
    """[8][N][K] float32, the law of sample_implicit_keywords_from_quantile_dfs on the singleton
    experiment quantiles (gymnasium_kw_utils.py:296-339, experiment_quantiles.py:16-25), vectorised."""
    rng = np.random.default_rng(seed)
    shape = (N, K)
    has = rng.random(shape) > no_vol_prob
    r = rng.random(shape)
    vol_mean = np.where(has, float(mean_volume), 0.0)
    vol_std = np.where(has, np.floor(1 + r * 0.5 * mean_volume), r * 0.5)
    loc = _pl(rng, 0.3, 0.55, 1.0, shape)
    scale = np.maximum(0.01, _pl(rng, 0.01, 0.15, 0.3, shape) * loc)
    bctr = _pl(rng, 0.1, 0.5, 0.9, shape)
    sctr = np.full(shape, cvr)
    mu = _pl(rng, 0.3, 1.0, 1.5, shape)
    sd = np.maximum(0.01, _pl(rng, 0.01, 0.15, 0.3, shape) * mu)
    return np.stack([vol_mean, vol_std, loc, scale, bctr, sctr, mu, sd]).astype(np.float32)


def explicit_params(N, K, seed):
    """[8][N][K] float32, the law of sample_random_keywords (gymnasium_kw_utils.py:129-140)"""
    rng = np.random.default_rng(seed)
    shape = (N, K)
    vm = (2 ** rng.beta(2, 5, size=shape) * 15 - 1).astype(int)
    vs = rng.random(size=shape) * 0.5 * (vm + 1)
    sctr = rng.beta(5, 2, size=shape)
    ic = rng.random(size=shape) * 1.5
    mu = rng.beta(2, 5, size=shape) * 1.5
    sd = rng.beta(2, 5, size=shape) * mu
    bctr = rng.beta(2, 5, size=shape)
    sl = rng.beta(5, 5, size=shape) * 25
    return np.stack([vm, vs, ic, sl, bctr, sctr, mu, sd]).astype(np.float32)


def mirror_oracle(engine, planes, **kw):
    """an OracleEngine holding the same parameters, keys and ticks as a StepEngine"""
    o = orc.OracleEngine(engine.num_envs, engine.num_keywords, model=engine.model, **kw)
    o.params[:] = planes
    k, t = engine.get_rng_state()
    o.key[:] = k
    o.tick[:] = t
    return o


def f32_dollars(cents):
    """what the kernels store: (float)cents / 100.0f, one correctly rounded f32 division"""
    return (np.asarray(cents).astype(np.float32) / np.float32(100.0)).astype(np.float32)


def assert_step_equal(got, ref, implicit=True):
    assert np.array_equal(got["impressions"], ref["impressions"])
    assert np.array_equal(got["buyside_clicks"], ref["clicks"])
    assert np.array_equal(got["sellside_conversions"], ref["conversions"])
    assert np.array_equal(got["revenue"], f32_dollars(ref["revenue_cents"]))
    if implicit:
        assert np.array_equal(got["cost"], f32_dollars(ref["cost_cents"]))
    else:
        assert np.array_equal(got["cost"], ref["cost"].astype(np.float32))
    assert np.array_equal(got["reward"], ref["reward"])            # f64, bit-exact
    assert np.array_equal(got["cumulative_profit"], ref["cum_profit"])
    assert np.array_equal(got["days_passed"], ref["day"])
    assert np.array_equal(got["terminated"], ref["terminated"])
    assert np.array_equal(got["truncated"], ref["truncated"])


def assert_outcome_lists(got, ref, K, env=0):
    """the combined BiddingOutcomes (bidding_simulation.py:10-38,124-147) as the reference recorded them in a fixture (`ref`:
    costs / revenues / revenues_per_cost per keyword, impression_share, profit) against regenerated ones (`got`: the same keys,
    lists indexed [env][keyword] or [keyword]): every list element for element and in order, impression_share the very same
    float64 (the lossy volume re-derivation of combine_outcomes included), profit within 1e-9 (its revenue sum is
    ndarray::sum in the reference's Rust, un-buildable here: the order of that one sum is not pinned)."""
    def row(x):
        return x[env] if len(x) != K or (K and isinstance(x[0], list) and x[0] and isinstance(x[0][0], list)) else x
    for name in ("costs", "revenues", "revenues_per_cost"):
        lists = row(got[name])
        assert len(lists) == K
        for k in range(K):
            assert [float(v) for v in lists[k]] == ref[name][k], (name, k)
    share = np.asarray(got["impression_share"], dtype=np.float64).reshape(-1, K)[env]
    assert share.tolist() == ref["impression_share"]
    np.testing.assert_allclose(np.asarray(got["profit"], dtype=np.float64).reshape(-1, K)[env], ref["profit"], rtol=0, atol=1e-9)
