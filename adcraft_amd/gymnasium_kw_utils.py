"""Host-side keyword generation and observation helpers - the engine-side mirror of
adcraft/gymnasium_kw_utils.py and adcraft/pull_quantiles_data/quantiles_to_keywords.py.

These run at reset() only (never on the step path).  The seeded draws reproduce the reference's
parameter tuples bit for bit: the DRAW ORDER on the shared numpy Generator is the contract
(SURVEY Appendix A.5), pinned by tests/golden/g2_*.json and the tuples printed in the reference's
notebooks.  The reference builds Python Keyword objects around these numbers; here the numbers are
uploaded to the device as struct-of-arrays planes (adc_param order).
"""
import numpy as np

from .spaces import get_action_space, get_observation_space  # noqa: F401  (reference exports them here)

# parameter tuple layouts (adcraft/gymnasium_kw_utils.py:20-28)
#   explicit: ((vol_mean, vol_std), imp_intercept, imp_slope, bctr, sctr, mean_revenue, std_revenue)
#   implicit: ((vol_mean, vol_std), cost_loc, 1/cost_scale, bctr, sctr, mean_revenue, std_revenue)   (printed form)


def sample_from_quantiles(n, num_buckets, mins, meds, maxs, rng):
    """quantiles_to_keywords.py:13-28.  `mins/meds/maxs` are indexed positionally here; the reference
    indexes a pandas Series by label, identical whenever the (filtered) frame keeps a 0..B-1 index."""
    out = []
    buckets = rng.integers(low=0, high=num_buckets, size=(n,))
    samples = rng.random(size=(n,))
    for bucket, q in zip(buckets, samples):
        out.append(np.interp(q, [0.0, 0.5, 1.0], [mins[bucket], meds[bucket], maxs[bucket]]))
    return out


def _column(data, name):
    col = data[name]
    return np.asarray(col.to_numpy() if hasattr(col, "to_numpy") else col, dtype=np.float64)


def sample_implicit_keyword_params(num_keywords, rng, keyword_config):
    """gymnasium_kw_utils.py:260-349 minus object construction.  Returns the list of
    (vol, cost_loc, cost_scale, bctr, sctr, rev_mean, rev_std) the reference passes to
    generate_implicit_keyword_from_params."""
    load = keyword_config.get("load_quant_func")
    make = keyword_config.get("make_quant_func")
    if load is None:
        raise NotImplementedError("keyword_config must provide load_quant_func (the reference's default reads "
                                  "user CSVs and its make_quantile_df_csvs raises NotImplementedError, "
                                  "gymnasium_kw_utils.py:229-231)")
    if keyword_config.get("quantiles_folder", False):
        data = load(keyword_config)
    else:
        if make is not None:
            make(keyword_config)
        data = load(keyword_config)
    assert data is not None, "Invalid quantile parameters specified in keyword_config for data"
    no_volume_prob = keyword_config.get("no_vol_prob", 0.0)
    nrows = len(_column(data, "min_vol"))
    lists = [[
        (int(v), int(1 + rng.random() * 0.5 * v)) if rng.random() > no_volume_prob and not np.isnan(v)
        else (0, rng.random() * 0.5)
        for v in sample_from_quantiles(num_keywords, nrows, _column(data, "min_vol"), _column(data, "median_vol"),
                                       _column(data, "max_vol"), rng)]]
    for param in ["ave_cpc", "std_cpc", "bctr", "sctr", "rpsc", "std_rpsc"]:
        keep = _column(data, f"count_{param}") > 0
        lists.append(sample_from_quantiles(num_keywords, int(keep.sum()), _column(data, f"min_{param}")[keep],
                                           _column(data, f"median_{param}")[keep], _column(data, f"max_{param}")[keep], rng))
        if param[:4] == "std_":      # un-normalise the standard deviations (:333-339)
            for i in range(num_keywords):
                lists[-1][i] = max([0.01, lists[-1][i] * lists[-2][i]])
    return [tuple(x) for x in zip(*lists)]


def sample_random_keyword_params(num_keywords, rng):
    """gymnasium_kw_utils.py:129-140 (ExplicitKeyword law of the default constructor)"""
    v_mean_list = (2 ** rng.beta(2, 5, size=num_keywords) * 15 - 1).astype(int)
    v_std_list = rng.random(size=num_keywords) * 0.5 * (v_mean_list + 1)
    vol_list = [(vm, vs) for vm, vs in zip(v_mean_list, v_std_list)]
    sctr_list = rng.beta(5, 2, size=num_keywords)
    imp_intercept_list = rng.random(size=num_keywords) * 1.5
    mean_revenue_list = rng.beta(2, 5, size=num_keywords) * 1.5
    std_revenue_list = rng.beta(2, 5, size=num_keywords) * mean_revenue_list
    bctr_list = rng.beta(2, 5, size=num_keywords)
    imp_slope_list = rng.beta(5, 5, size=num_keywords) * 25
    return list(zip(vol_list, imp_intercept_list, imp_slope_list, bctr_list, sctr_list, mean_revenue_list,
                    std_revenue_list))


def consume_construction_draws(num_keywords, rng):
    """The reference validates every keyword's reward sampler at construction with rds(2), rds(5),
    rds(5) (adcraft/synthetic_kw_classes.py:337-339): 12 normal draws per keyword on the env
    generator, after the parameters and before anything else.  Replayed so that later draws on
    np_random (e.g. a user calling env.np_random) line up with the reference."""
    for _ in range(num_keywords):
        rng.normal(0.0, 1.0, 2)
        rng.normal(0.0, 1.0, 5)
        rng.normal(0.0, 1.0, 5)


def implicit_params_to_planes(params):
    """[(vol, loc, scale, bctr, sctr, mu, sd)] -> float32 [8][K] in adc_param order"""
    K = len(params)
    out = np.zeros((8, K), dtype=np.float32)
    for k, (vol, loc, scale, bctr, sctr, mu, sd) in enumerate(params):
        out[:, k] = (vol[0], vol[1], loc, scale, bctr, sctr, mu, sd)
    return out


def explicit_params_to_planes(params):
    """[(vol, intercept, slope, bctr, sctr, mu, sd)] -> float32 [8][K]"""
    K = len(params)
    out = np.zeros((8, K), dtype=np.float32)
    for k, (vol, ic, sl, bctr, sctr, mu, sd) in enumerate(params):
        out[:, k] = (vol[0], vol[1], ic, sl, bctr, sctr, mu, sd)
    return out


def printed_params(params, implicit):
    """the keyword_params lists the reference keeps: mutable [vol, a, b, bctr, sctr, mu, sd];
    implicit keywords show 1/cost_scale in slot 2 (gymnasium_kw_utils.py:195)"""
    out = []
    for p in params:
        p = list(p)
        if implicit:
            p[2] = 1 / p[2]
        out.append(p)
    return out


def repr_params(params):
    """gymnasium_kw_utils.py:352-370"""
    return ",   ".join(name + f": {value}" for name, value in zip(
        ["volume", "imp_intercept", "imp_slope", "bctr", "sctr", "mean revenue", "std revenue"], params))


def repr_all_params(params_list):
    """gymnasium_kw_utils.py:373-380"""
    return "\n".join(f"kw{n} params:\n {repr_params(params)}" for n, params in enumerate(params_list))


def flatten_dict_array(obs):
    """gymnasium_kw_utils.py:383-390: sorted-key concatenation"""
    return np.hstack([np.asarray(obs[k]).ravel() for k in sorted(obs.keys())])


FLAT_OBS_KEYS = ("buyside_clicks", "cost", "cumulative_profit", "days_passed", "impressions", "revenue",
                 "sellside_conversions")          # sorted(): the layout FlatArrayWrapper emits


# ---- the reference's experiment quantile tables (experiment_utils/experiment_quantiles.py:16-47) ----
def generate_simple_experiment_quantiles(mean_volume, cvr):
    """dict-of-columns equivalent of the singleton quantile DataFrame (one bucket per parameter)"""
    d = {"vol": [mean_volume] * 3, "ave_cpc": [0.3, 0.55, 1], "std_cpc": [0.01, 0.15, 0.3], "bctr": [0.1, 0.5, 0.9],
         "sctr": [cvr] * 3, "rpsc": [0.3, 1.0, 1.5], "std_rpsc": [0.01, 0.15, 0.3]}
    table = {}
    for k, v in d.items():
        table[f"count_{k}"] = [3]
        table[f"min_{k}"] = [v[0]]
        table[f"median_{k}"] = [v[1]]
        table[f"max_{k}"] = [v[2]]
    return table


def experiment_keyword_config(mean_volume, conversion_rate, **extra):
    """keyword_config equivalent to the reference's experiment configs (experiment_configs.py:15-98) without
    the CSV round trip: load_quant_func returns the in-memory table."""
    table = generate_simple_experiment_quantiles(mean_volume, conversion_rate)
    cfg = {"quantiles_folder": "in-memory", "load_quant_func": lambda kc: table,
           "mean_volume": mean_volume, "conversion_rate": conversion_rate}
    cfg.update(extra)
    return cfg
