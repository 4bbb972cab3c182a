#!/usr/bin/env python3
"""G10: traces of the REFERENCE's NaiveZeroMarginStrategy (adcraft/baselines/interpolated_expectations.py:442-515),
imported unmodified, driven exactly as the notebooks drive it (run_heatmap_experiments.ipynb cell 1:
update_all_caches -> cache_tensors_to_floats -> sample_action -> env.step with the agent's bids and a fixed budget).

The campaign itself is the reference's simulate_epoch_of_bidding_on_campaign on keywords from its quantile sampler
(same in-memory stand-ins as tools/gen_golden.py; the Rust volume sampler replaced by its law on a seeded numpy
generator).  The agent's rng is wrapped so that every uniform it draws - and for which keyword - is recorded.

Stored per step: the observation the agent was given (as the float32 values torch.Tensor(...) makes of it), the
uniforms it drew (NaN where it drew none), the action it returned (float64 bids, budget) and its caches afterwards.

Usage: python tools/gen_golden_agent.py      (rewrites tests/golden/g10_zero_margin_agent.json)
"""
import json
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

CASES = [
    dict(seed=5, agent_seed=0, K=8, T=40, mean_volume=64, cvr=0.8, default_rpc=1.0),
    dict(seed=6, agent_seed=1, K=10, T=40, mean_volume=16, cvr=0.3, default_rpc=1.0),
    dict(seed=7, agent_seed=2, K=6, T=60, mean_volume=4, cvr=0.1, default_rpc=3.0),
    dict(seed=8, agent_seed=3, K=5, T=30, mean_volume=128, cvr=1.0, default_rpc=1.0),
]


class RecordingRng:
    """stands where agent.rng stands; sample_action only ever calls .random()"""

    def __init__(self, rng):
        self.rng, self.draws = rng, []

    def random(self):
        u = float(self.rng.random())
        self.draws.append(u)
        return u


def cache_tensors_to_floats(cache):          # run_heatmap_experiments.ipynb cell 1, verbatim behaviour
    cache["ave_rpc"] = float(cache["ave_rpc"])
    cache["ave_sctr"] = float(cache["ave_sctr"])
    cache["ave_clicks"] = {k: [float(v[0]), v[1]] for k, v in cache["ave_clicks"].items()}


def main():
    warnings.filterwarnings("ignore")
    G.install_standins()
    from adcraft import bidding_simulation as b, gymnasium_kw_utils as u
    from adcraft.experiment_utils import experiment_quantiles as eq
    import adcraft.baselines.interpolated_expectations as ie
    out = []
    for cs in CASES:
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(cs["seed"])))
        cfg, _ = G.quant_cfg(eq, cs["mean_volume"], cs["cvr"], None)
        _, params = u.sample_implicit_keywords_from_quantile_dfs(cs["K"], rng, cfg)
        kws = []
        for p in params:
            vol = (float(p[0][0]), float(p[0][1]))
            kw, _ = u.generate_implicit_keyword_from_params(vol, G.f32x(p[1]), G.f32x(1.0 / p[2]), G.f32x(p[3]), G.f32x(p[4]),
                                                            G.f32x(p[5]), G.f32x(p[6]), rng)
            kw.volume_sampler = (lambda m, s: (lambda: int(np.floor(max(rng.normal(m, max(s, 1e-12)), 0.0) + 0.5))))(vol[0], vol[1])
            kws.append(kw)
        K = cs["K"]
        agent = ie.NaiveZeroMarginStrategy(K, default_expected_revenue_per_conversion=cs["default_rpc"], seed=cs["agent_seed"])
        rec = RecordingRng(agent.rng)
        agent.rng = rec
        obs = {k: np.zeros(K) for k in ("impressions", "buyside_clicks", "cost", "sellside_conversions", "revenue")}
        action = {"budget": 0.0, "keyword_bids": 0.01 + np.zeros((K,))}
        steps = []
        for t in range(cs["T"]):
            agent.update_all_caches(action, obs)
            for c in agent.caches:
                cache_tensors_to_floats(c)
            # which keywords draw: exactly those with num_rpc_obs < 1, in keyword order (sample_action, :500-501)
            draws_for = [i for i in range(K) if agent.caches[i]["num_rpc_obs"] < 1]
            rec.draws = []
            action = agent.sample_action()
            assert len(rec.draws) == len(draws_for)
            uni = [float("nan")] * K
            for i, d in zip(draws_for, rec.draws):
                uni[i] = d
            steps.append(dict(
                obs_clicks=[float(np.float32(x)) for x in obs["buyside_clicks"]],
                obs_conversions=[float(np.float32(x)) for x in obs["sellside_conversions"]],
                obs_revenue=[float(np.float32(x)) for x in obs["revenue"]],
                uniforms=uni,
                bids=[float(x) for x in action["keyword_bids"]], budget=float(action["budget"]),
                ave_rpc=[float(c["ave_rpc"]) for c in agent.caches], num_rpc_obs=[int(c["num_rpc_obs"]) for c in agent.caches],
                ave_sctr=[float(c["ave_sctr"]) for c in agent.caches], num_sctr_obs=[float(c["num_sctr_obs"]) for c in agent.caches],
                max_bids=[float(x) for x in agent.max_bids]))
            oc = b.simulate_epoch_of_bidding_on_campaign(kws, [float(x) for x in np.round(action["keyword_bids"], 2)], 100000)
            obs = dict(
                impressions=np.array([o["impressions"] for o in oc]),
                buyside_clicks=np.array([o["buyside_clicks"] for o in oc]),
                sellside_conversions=np.array([o["sellside_conversions"] for o in oc]),
                cost=np.array([float(np.sum(o["costs"])) if len(o["costs"]) else 0.0 for o in oc]),
                revenue=np.array([float(np.sum(o["revenues"])) if len(o["revenues"]) else 0.0 for o in oc]))
        out.append(dict(K=K, T=cs["T"], default_rpc=cs["default_rpc"], agent_seed=cs["agent_seed"], steps=steps))
        print("case", cs["seed"], "done:", sum(np.isfinite(s["uniforms"]).sum() for s in steps), "draws")
    path = os.path.join(G.OUT, "g10_zero_margin_agent.json")
    with open(path, "w") as f:
        json.dump(dict(source="adcraft/baselines/interpolated_expectations.py:442-515 (NaiveZeroMarginStrategy), executed "
                              "unmodified inside the loop of run_heatmap_experiments.ipynb cell 1", cases=out), f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
