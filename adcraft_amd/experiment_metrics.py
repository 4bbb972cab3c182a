"""AKNCP / NCP and ideal-profit helpers - mirror of adcraft/experiment_utils/experiment_metrics.py.

Host numpy versions of the episode-level reductions (they run once per episode on [T x K]
matrices).  For sharded runs the per-keyword sums come from the engine's device accumulators and one
all-reduce (adcraft_amd/distributed.py).
"""
import numpy as np


def get_implicit_kw_bid_cpc_impressions(implicit_keyword, bid_array, n_samples=2048):
    """experiment_metrics.py:20-37"""
    second_prices = np.reshape(np.sort(implicit_keyword.sample_bids(n_samples)), (-1,))
    indices = np.searchsorted(second_prices, bid_array, side="right")
    impression_rates = indices / n_samples
    indices = np.minimum(indices, n_samples - 1)
    mean_prices = np.cumsum(second_prices) / np.arange(1, n_samples + 1, 1)
    return impression_rates, mean_prices[indices]


def get_max_expected_bid_profits(kw_params, expected_cpc_per_bid, expected_impression_rate_per_bid):
    """experiment_metrics.py:40-61"""
    expected_profits = np.maximum(
        kw_params[0][0] * expected_impression_rate_per_bid * kw_params[3]
        * (kw_params[4] * kw_params[5] - expected_cpc_per_bid), 0.0)
    return (max([0.0, expected_profits.max()]), np.sum(expected_profits > 0) / len(expected_cpc_per_bid),
            np.argmax(expected_profits))


def compute_AKNCP(kw_profits, ideal_profits):
    """experiment_metrics.py:64-75"""
    denominator = ideal_profits.copy()
    denominator[denominator <= 0] = 1.0
    denominator = denominator.mean(axis=0)
    return np.median(kw_profits.mean(axis=0) / denominator)


def compute_NCP(kw_profits, ideal_profits):
    """experiment_metrics.py:78-83"""
    denominator = ideal_profits.sum()
    if denominator <= 0.0:
        denominator = 1.0
    return kw_profits.sum() / denominator


def akncp_ncp_from_sums(sum_profit_k, sum_ideal_k, sum_ideal_pos_k=None):
    """AKNCP / NCP from per-keyword sums over (time x envs), the form the sharded engine reduces to.
    sum_ideal_pos_k: sum of ideal' (ideal with <=0 -> 1), defaults to the same replacement on the sums."""
    sum_profit_k = np.asarray(sum_profit_k, dtype=np.float64)
    sum_ideal_k = np.asarray(sum_ideal_k, dtype=np.float64)
    den = np.asarray(sum_ideal_pos_k, dtype=np.float64) if sum_ideal_pos_k is not None else np.where(sum_ideal_k <= 0, 1.0, sum_ideal_k)
    total = sum_ideal_k.sum()
    return float(np.median(sum_profit_k / den)), float(sum_profit_k.sum() / (total if total > 0 else 1.0))
