#!/usr/bin/env python3
"""G11: cells of the paper's heat-map experiment run by the REFERENCE itself, end to end
(adcraft/baseline_experiment_and_figs_notebooks/run_heatmap_experiments.ipynb cells 1-4): the reference's
BiddingSimulation env (unmodified, in-memory stand-ins as tools/gen_golden.py; the one Rust sampler on the path - the
volume draw, src/lib.rs:314-325 - replaced by its law on a seeded numpy generator), its NaiveZeroMarginStrategy, its
get_implicit_kw_bid_cpc_impressions / get_max_expected_bid_profits / compute_AKNCP / compute_NCP, driven by the
notebook's run_zero_margin_agent loop (restated below line for line; it lives in a notebook cell, not in a module).

Stored per cell: the per-run AKNCP / NCP / total profit and the keyword parameters of each env seed.  The engine's
device-resident loop must reproduce the cell means within the run-to-run noise (tests/test_gpu_policies.py).

Usage: python tools/gen_golden_heatmap.py     (rewrites tests/golden/g11_heatmap_cells.json; about 10 minutes)
"""
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

# drift=True: the non-stationary experiment (timing_and_other_one_off_experiments.ipynb cell 3: updater_mask all True)
CELLS = [dict(mean_volume=16, cvr=0.67, drift=False), dict(mean_volume=128, cvr=0.34, drift=False),
         dict(mean_volume=4, cvr=1.0, drift=False), dict(mean_volume=16, cvr=0.67, drift=True)]
ENV_SEEDS, AGENT_SEEDS = (5, 6, 7, 8), (0, 1, 2, 3)
K, DAYS = 100, 60


def cache_tensors_to_floats(cache):          # notebook cell 1
    cache["ave_rpc"] = float(cache["ave_rpc"])
    cache["ave_sctr"] = float(cache["ave_sctr"])
    cache["ave_clicks"] = {k: [float(v[0]), v[1]] for k, v in cache["ave_clicks"].items()}


def run_zero_margin_agent(agent, env, irs, cpcs, m, budget=100000):
    """notebook cell 1 (after env.reset(seed) by the caller, as in cell 3)"""
    previous_observation, info = env.reset()
    previous_action = {"budget": 0.0, "keyword_bids": 0.01 + np.zeros((env.num_keywords,))}
    truncated, terminated = False, False
    kw_profits, ideal_profits = [], []
    while not (terminated or truncated):
        agent.update_all_caches(previous_action, previous_observation)
        for i in range(env.num_keywords):
            cache_tensors_to_floats(agent.caches[i])
        action = agent.sample_action()
        ideal_profit = []
        for kw_index, kw_params in enumerate(env.keyword_params):
            max_exp_profit, _, _ = m.get_max_expected_bid_profits(kw_params, cpcs[kw_index], irs[kw_index])
            ideal_profit.append(max_exp_profit)
        ideal_profits.append(ideal_profit)
        previous_observation, reward, terminated, truncated, info = env.step(
            action={"budget": budget, "keyword_bids": action["keyword_bids"]})
        previous_action = action
        kw_profits.append(previous_observation["revenue"] - previous_observation["cost"])
    return np.array(kw_profits), np.array(ideal_profits)


def run_oracle_agent(env, irs, cpcs, m, allowed_bids, budget=100000):
    """timing_and_other_one_off_experiments.ipynb cell 2 (after env.reset(seed) by the caller)"""
    previous_observation, info = env.reset()
    truncated, terminated = False, False
    kw_profits, ideal_profits = [], []
    while not (terminated or truncated):
        ideal_profit, kw_actions = [], []
        for kw_index, kw_params in enumerate(env.keyword_params):
            max_exp_profit, _, optimal_bid_index = m.get_max_expected_bid_profits(kw_params, cpcs[kw_index], irs[kw_index])
            kw_actions.append(allowed_bids[optimal_bid_index])
            ideal_profit.append(max_exp_profit)
        ideal_profits.append(ideal_profit)
        previous_observation, reward, terminated, truncated, info = env.step(
            action={"budget": budget, "keyword_bids": np.array(kw_actions)})
        kw_profits.append(previous_observation["revenue"] - previous_observation["cost"])
    return np.array(kw_profits), np.array(ideal_profits)


def main():
    warnings.filterwarnings("ignore")
    rust = G.install_standins()
    vol_rng = np.random.default_rng(12345)
    rust.nonneg_int_normal_sampler = lambda mean, std: int(np.floor(max(vol_rng.normal(mean, max(std, 1e-12)), 0.0) + 0.5))
    from adcraft import gymnasium_kw_env as kw_sim
    from adcraft.experiment_utils import experiment_metrics as m, experiment_quantiles as eq
    import adcraft.baselines.interpolated_expectations as ie
    out = []
    for cell in CELLS:
        cfg, _ = G.quant_cfg(eq, cell["mean_volume"], cell["cvr"])

        def make_env():
            return kw_sim.bidding_sim_creator(env_config=dict(
                keyword_config=cfg, num_keywords=K, max_days=DAYS,
                updater_params=[["vol", 0.03], ["ctr", 0.03], ["cvr", 0.03]], updater_mask=[True] * K if cell["drift"] else None))
        env = make_env()
        runs, params_by_seed = [], {}
        t0 = time.time()
        for env_seed in ENV_SEEDS:
            for agent_seed in AGENT_SEEDS:
                if cell["drift"]:
                    env = make_env()         # as get_nonstationary_profits does (init_volumes belong to this keyword set)
                env.reset(seed=env_seed)
                params_by_seed[str(env_seed)] = G.params_to_json(env.keyword_params)
                allowed_bids = np.arange(0.01, 3.00, 0.01)
                irs, cpcs = [], []
                for kw in env.keywords:
                    ir, cpc = m.get_implicit_kw_bid_cpc_impressions(kw, allowed_bids)
                    irs.append(ir)
                    cpcs.append(cpc)
                agent = ie.NaiveZeroMarginStrategy(env.num_keywords, default_expected_revenue_per_conversion=1.0, seed=agent_seed)
                kw_profits, ideal_profits = run_zero_margin_agent(agent, env, irs, cpcs, m)
                runs.append(dict(env_seed=env_seed, agent_seed=agent_seed, AKNCP=float(m.compute_AKNCP(kw_profits, ideal_profits)),
                                 NCP=float(m.compute_NCP(kw_profits, ideal_profits)), total_profit=float(kw_profits.sum()),
                                 total_ideal=float(ideal_profits.sum()), days=int(kw_profits.shape[0])))
                print(cell, runs[-1], f"{time.time() - t0:.0f} s", flush=True)
        oracle_runs = []
        for env_seed in ENV_SEEDS:
            for rep in range(2):
                if cell["drift"]:
                    env = make_env()
                env.reset(seed=env_seed)
                allowed_bids = np.arange(0.01, 3.01, 0.01)               # the oracle notebook's grid (300 points)
                irs, cpcs = [], []
                for kw in env.keywords:
                    ir, cpc = m.get_implicit_kw_bid_cpc_impressions(kw, allowed_bids)
                    irs.append(ir)
                    cpcs.append(cpc)
                kw_profits, ideal_profits = run_oracle_agent(env, irs, cpcs, m, allowed_bids)
                oracle_runs.append(dict(env_seed=env_seed, AKNCP=float(m.compute_AKNCP(kw_profits, ideal_profits)),
                                        NCP=float(m.compute_NCP(kw_profits, ideal_profits)), total_profit=float(kw_profits.sum()),
                                        total_ideal=float(ideal_profits.sum()), days=int(kw_profits.shape[0])))
                print(cell, "oracle", oracle_runs[-1], f"{time.time() - t0:.0f} s", flush=True)
        out.append(dict(cell, K=K, days=DAYS, runs=runs, oracle_runs=oracle_runs, keyword_params=params_by_seed))
    path = os.path.join(G.OUT, "g11_heatmap_cells.json")
    with open(path, "w") as f:
        json.dump(dict(source="run_heatmap_experiments.ipynb cells 1-4 on the reference's own env, agent and metrics "
                              "(stationary keywords: updater_mask=None, gymnasium_kw_env.py:127-128; drift cells: updater_mask all True, "
                              "timing_and_other_one_off_experiments.ipynb cell 3)", cells=out), f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
