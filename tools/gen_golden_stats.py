#!/usr/bin/env python3
"""G9: distribution of whole-step outcomes of the REFERENCE (its unmodified Python loop) on fixed keyword sets.

Same in-memory stand-ins as tools/gen_golden.py; the only Rust sampler on this path (the volume draw,
src/lib.rs:314-325) is replaced by its law round_half_away(max(N(mean, std), 0)) on a seeded numpy generator.
For each scenario the reference's simulate_epoch_of_bidding_on_campaign is run T times with fixed bids and budget;
per-keyword mean / variance of impressions, clicks, cost, conversions, revenue and of the step reward are stored.
tests compare the engine's own stream against these moments (z-test), including a binding-budget scenario.

Usage: python tools/gen_golden_stats.py     (rewrites tests/golden/g9_step_statistics.json; takes a few minutes)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

T = 4000
SCENARIOS = [
    dict(name="dense_nonbinding", seed=101, K=6, mean_volume=64, cvr=0.8, no_vol_prob=None, budget=1.0e9, bid=(0.4, 1.0)),
    dict(name="dense_binding", seed=102, K=6, mean_volume=64, cvr=0.8, no_vol_prob=None, budget=25.0, bid=(0.5, 1.1)),
    dict(name="sparse", seed=103, K=8, mean_volume=16, cvr=0.1, no_vol_prob=0.5, budget=1.0e9, bid=(0.3, 1.2)),
    dict(name="tight_budget", seed=104, K=5, mean_volume=40, cvr=0.5, no_vol_prob=None, budget=3.0, bid=(0.6, 1.2)),
]


def main():
    G.install_standins()
    from adcraft import bidding_simulation as b, gymnasium_kw_utils as u
    from adcraft.experiment_utils import experiment_quantiles as eq
    out = []
    for sc in SCENARIOS:
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(sc["seed"])))
        cfg, _ = G.quant_cfg(eq, sc["mean_volume"], sc["cvr"], sc["no_vol_prob"])
        _, params = u.sample_implicit_keywords_from_quantile_dfs(sc["K"], rng, cfg)
        kws, kp = [], []
        for p in params:
            vol = (float(p[0][0]), float(p[0][1]))
            loc, scale = G.f32x(p[1]), G.f32x(1.0 / p[2])
            bctr, sctr, mu, sd = G.f32x(p[3]), G.f32x(p[4]), G.f32x(p[5]), G.f32x(p[6])
            kw, _ = u.generate_implicit_keyword_from_params(vol, loc, scale, bctr, sctr, mu, sd, rng)
            kw.volume_sampler = (lambda m, s: (lambda: int(np.floor(max(rng.normal(m, max(s, 1e-12)), 0.0) + 0.5))))(vol[0], vol[1])
            kws.append(kw)
            kp.append(dict(vol_mean=vol[0], vol_std=vol[1], loc=loc, scale=scale, bctr=bctr, sctr=sctr, rev_mean=mu, rev_std=sd))
        bids = [float(x) for x in np.around(rng.uniform(sc["bid"][0], sc["bid"][1], sc["K"]), 2)]
        acc = {k: np.zeros((T, sc["K"])) for k in ("impressions", "buyside_clicks", "cost", "sellside_conversions", "revenue")}
        reward = np.zeros(T)
        for t in range(T):
            oc = b.simulate_epoch_of_bidding_on_campaign(kws, bids, sc["budget"])
            for k, o in enumerate(oc):
                acc["impressions"][t, k] = o["impressions"]
                acc["buyside_clicks"][t, k] = o["buyside_clicks"]
                acc["sellside_conversions"][t, k] = o["sellside_conversions"]
                acc["cost"][t, k] = float(np.sum(o["costs"])) if len(o["costs"]) else 0.0
                acc["revenue"][t, k] = float(np.sum(o["revenues"])) if len(o["revenues"]) else 0.0
            reward[t] = sum(o["profit"] for o in oc)
            if t % 500 == 0:
                print(sc["name"], t, flush=True)
        out.append(dict(name=sc["name"], K=sc["K"], budget=sc["budget"], bids=bids, keyword_params=kp, steps=T,
                        mean={k: v.mean(axis=0).tolist() for k, v in acc.items()},
                        var={k: v.var(axis=0, ddof=1).tolist() for k, v in acc.items()},
                        reward_mean=float(reward.mean()), reward_var=float(reward.var(ddof=1)),
                        frac_budget_exhausted=float((acc["cost"].sum(axis=1) > sc["budget"] - 1.5).mean())))
    G.dump("g9_step_statistics.json", dict(
        source="adcraft/bidding_simulation.py:170-234 executed unmodified, T independent days per scenario with fixed bids; "
               "volume sampler = the law of src/lib.rs:314-325 on a seeded numpy generator; moments over the T days",
        scenarios=out))


if __name__ == "__main__":
    main()
