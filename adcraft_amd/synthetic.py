"""Synthetic keyword sets for benchmarks: the BASELINE configs' parameter laws, vectorised over
[num_envs, num_keywords] (host-side numpy; runs once, before any timed region).

Law = adcraft/gymnasium_kw_utils.py:296-339 (sample_implicit_keywords_from_quantile_dfs) on the singleton
experiment quantiles of adcraft/experiment_utils/experiment_quantiles.py:16-47.  The per-env seeded,
draw-order-exact generator used by reset(seed=...) lives in gymnasium_kw_utils.py of this package.
"""
import numpy as np

CONFIGS = {
    # name: (num_envs, num_keywords, mean_volume, cvr, no_vol_prob, drift)   BASELINE.json configs[1..4]
    "cfg2": (4096, 256, 128, 0.8, 0.0, False),     # dense stationary (experiment_configs.py:15-27)
    "cfg3": (16384, 1024, 16, 0.1, 0.5, False),    # sparse volume (experiment_configs.py:57-69 + no_vol_prob)
    "cfg4": (8192, 1024, 128, 0.8, 0.0, False),    # per-GPU shard of 65536 x 1024
    "cfg5": (2048, 1024, 128, 0.8, 0.0, True),     # per-GPU shard of 16384 x 1024, drift on (experiment_configs.py:72-84)
}


def _pl(rng, lo, mid, hi, shape):
    return np.interp(rng.random(shape), [0.0, 0.5, 1.0], [lo, mid, hi])


def implicit_keyword_planes(num_envs, num_keywords, seed, mean_volume=128, cvr=0.8, no_vol_prob=0.0):
    """float32 [8][N][K] in adc_param order"""
    rng = np.random.default_rng(seed)
    shape = (num_envs, num_keywords)
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


def explicit_keyword_planes(num_envs, num_keywords, seed):
    """float32 [8][N][K] in adc_param order: the law of sample_random_keywords (adcraft/gymnasium_kw_utils.py:129-140: the
    default-constructor, EXPLICIT keyword set), its eight draws in its order, for all envs at once"""
    rng = np.random.default_rng(seed)
    shape = (num_envs, num_keywords)
    vol_mean = (2 ** rng.beta(2, 5, size=shape) * 15 - 1).astype(int)
    vol_std = rng.random(size=shape) * 0.5 * (vol_mean + 1)
    sctr = rng.beta(5, 2, size=shape)
    intercept = rng.random(size=shape) * 1.5
    rev_mean = rng.beta(2, 5, size=shape) * 1.5
    rev_std = rng.beta(2, 5, size=shape) * rev_mean
    bctr = rng.beta(2, 5, size=shape)
    slope = rng.beta(5, 5, size=shape) * 25
    return np.stack([vol_mean, vol_std, intercept, slope, bctr, sctr, rev_mean, rev_std]).astype(np.float32)
