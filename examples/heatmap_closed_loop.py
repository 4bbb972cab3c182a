#!/usr/bin/env python3
"""The volume x conversion-rate heat-map experiment of the paper
(adcraft/baseline_experiment_and_figs_notebooks/run_heatmap_experiments.ipynb cells 1-4), device-resident.

The notebook runs, for every (mean_volume, conversion_rate) cell, env seeds 5..8 x agent seeds 0..3 one after the
other: reset(seed), sample the bid curves of the 100 keywords, then 60 days of NaiveZeroMarginStrategy (stationary
keywords: the notebook passes updater_mask=None, for which update_keywords returns at once, gymnasium_kw_env.py:127-128), and
stores kw_profits / ideal_profits.  Here the 16 (env seed, agent seed) runs of a cell are the 16 envs of ONE engine and
a day is three kernel launches (agent, ideal profit, step) for all of them.

Usage: python examples/heatmap_closed_loop.py [--volumes 1 16 256] [--cvrs 0.1 0.5 1.0] [--policy zero_margin|oracle]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import gymnasium_kw_utils as utils  # noqa: E402
from adcraft_amd.closed_loop import run_baseline_episode  # noqa: E402
from adcraft_amd.engine import StepEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volumes", type=float, nargs="+", default=[2.0 ** k for k in range(0, 11, 2)])
    ap.add_argument("--cvrs", type=float, nargs="+", default=[0.01, 0.34, 0.67, 1.0])
    ap.add_argument("--policy", default="zero_margin")
    ap.add_argument("--num-keywords", type=int, default=100)
    ap.add_argument("--days", type=int, default=60)
    args = ap.parse_args()
    env_seeds, agent_seeds = range(5, 9), range(0, 4)                   # notebook: range(e_s, 9) x range(a_s, 4)
    runs = [(es, ag) for es in env_seeds for ag in agent_seeds]
    N, K = len(runs), args.num_keywords
    t0 = time.perf_counter()
    print(f"{'volume':>8} {'cvr':>6} {'AKNCP':>8} {'NCP':>8}   (mean over {N} runs of {args.days} days, K = {K})")
    for vol in args.volumes:
        for cvr in args.cvrs:
            cfg = utils.experiment_keyword_config(vol, cvr)
            planes = np.zeros((8, N, K), np.float32)
            for i, (es, _) in enumerate(runs):                           # env.reset(seed=env_seed): the reference's draws
                rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(es)))
                planes[:, i] = utils.implicit_params_to_planes(utils.sample_implicit_keyword_params(K, rng, cfg))
            eng = StepEngine(N, K, max_days=args.days, loss_threshold=10000.0, drift_enabled=False)
            eng.set_all_params(planes)
            eng.reset(seeds=np.array([1000 * es + ag for es, ag in runs], dtype=np.uint64))
            r = run_baseline_episode(eng, args.policy, steps=args.days, budget=100000.0, default_rpc=1.0,
                                     agent_seeds=np.array([ag for _, ag in runs], dtype=np.uint64))
            eng.close()
            print(f"{vol:8.0f} {cvr:6.2f} {np.mean(r['AKNCP']):8.3f} {np.mean(r['NCP']):8.3f}")
    print(f"{len(args.volumes) * len(args.cvrs)} cells x {N} runs x {args.days} days in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
