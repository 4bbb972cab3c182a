"""The paper's baseline experiments as one device-resident loop (run_heatmap_experiments.ipynb cell 1,
timing_and_other_one_off_experiments.ipynb cell 2): per step, for all envs at once and without leaving the GPU,

    agent.update_all_caches(previous action, previous observation); action = agent.sample_action()
    ideal[k] = get_max_expected_bid_profits(current keyword params, cpc[k], ir[k])
    observation = env.step({"budget": budget, "keyword_bids": action["keyword_bids"]})
    kw_profits.append(observation["revenue"] - observation["cost"])

and at the end compute_AKNCP / compute_NCP per env from the accumulated sums.
"""
import numpy as np

from . import experiment_metrics as em


def run_baseline_episode(engine, policy="zero_margin", steps=None, budget=100000.0, default_rpc=1.0, agent_seeds=None,
                         n_samples=2048, bid_grid=None, curves=True, per_keyword_sums=True):
    """One episode of `steps` days (default: the engine's max_days) for every env of `engine` (already reset, keywords
    set).  policy: "zero_margin" (NaiveZeroMarginStrategy) or "oracle" (bid the argmax of the expected profit).
    Returns dict(kw_profit_sum [N, K], ideal_sum [N, K], AKNCP [N], NCP [N]).  per_keyword_sums=False: only AKNCP and NCP,
    reduced on the device (the per-env median over the keywords included): 2 N numbers cross the bus, not 3 N K."""
    if policy not in ("zero_margin", "oracle"):
        raise ValueError("policy must be 'zero_margin' or 'oracle'")
    steps = int(engine.max_days if steps is None else steps)
    if curves:
        engine.bid_curves_build(n_samples, bid_grid)           # irs, cpcs = get_implicit_kw_bid_cpc_impressions(...) per keyword
    engine.metrics_enable(True)
    engine.metrics_reset()
    if policy == "zero_margin":
        engine.agent_init(default_rpc, agent_seeds)
    engine.run_days(policy, steps, budget)     # agent / ideal profit / step for every day, one host call
    n = float(steps)
    if not per_keyword_sums:
        akncp, ncp = engine.metrics_akncp_ncp(n)
        return dict(AKNCP=akncp, NCP=ncp)
    profit, ideal, ideal_pos = engine.metrics_read_nk()
    with np.errstate(divide="ignore", invalid="ignore"):
        akncp = np.median((profit / n) / (ideal_pos / n), axis=1)             # experiment_metrics.py:64-76
    den = ideal.sum(axis=1)
    ncp = profit.sum(axis=1) / np.where(den <= 0.0, 1.0, den)                 # :79-83
    return dict(kw_profit_sum=profit, ideal_sum=ideal, ideal_pos_sum=ideal_pos, AKNCP=akncp, NCP=ncp)


__all__ = ["run_baseline_episode", "em"]
