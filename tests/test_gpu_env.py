"""-m gpu: the drop-in Python surface (BiddingSimulation, vector env, adaptors) on a real device.
The first block mirrors the reference's own env tests (adcraft/tests/test_env.py:10-69)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import adcraft_amd
    from adcraft_amd import _ffi
    assert _ffi.device_count() >= 1
    return adcraft_amd


def _cfg(mv=128, cvr=0.8, **kw):
    from adcraft_amd import gymnasium_kw_utils as utils
    return utils.experiment_keyword_config(mv, cvr, **kw)


# ---------------------------------------------------------------- adcraft/tests/test_env.py
def test_reference_env_tests(pkg):
    env = pkg.BiddingSimulation()
    try:
        import gymnasium as gym
        assert isinstance(env, gym.Env)
    except ImportError:
        pass
    assert hasattr(env, "observation_space") and hasattr(env, "action_space")
    for s in (None, 1):
        env.reset(seed=s)
    reset_obs, reset_info = env.reset()
    assert env.observation_space.contains(reset_obs)
    assert "keyword_params" in reset_info
    action = env.action_space.sample()
    next_obs, reward, done, truncated, info = env.step(action)
    for k, v in reset_obs.items():
        next_obs[k] = next_obs[k].astype(v.dtype)
    assert env.observation_space.contains(next_obs)
    assert isinstance(reward, float) and isinstance(done, bool) and isinstance(truncated, bool)
    assert set(info) == {"bids", "bidding_outcomes", "keyword_params"}
    assert str(info["bidding_outcomes"]).startswith("[{'bid':") and "kw0 params" in str(info["keyword_params"])
    env.close()


@pytest.mark.parametrize("implicit,budget", [(True, 1.0e6), (True, 3.0), (False, 1.0e6), (False, 40.0)])
def test_bidding_outcomes_lists_every_click_on_demand(pkg, implicit, budget):
    """info["bidding_outcomes"] (src/lib.rs:251-275): the per-click lists are regenerated from the stream when the string is
    read - every cost and revenue of the step, consistent with the step's observations to the cent, under binding and
    non-binding budgets, for both keyword models; and a string read only after the NEXT step is still the first step's."""
    import ast

    def make():
        kw = dict(keyword_config=_cfg(40, 0.5), num_keywords=24) if implicit else dict(num_keywords=24)
        e = pkg.BiddingSimulation(budget=budget, **kw)
        e.reset(seed=5)
        return e

    env, twin = make(), make()
    rng = np.random.default_rng(2)
    bound = False
    kept = []
    for day in range(3):
        act = {"keyword_bids": rng.uniform(0.4, 1.6, 24).round(2), "budget": np.array([budget])}
        obs, reward, *_, info = env.step(act)
        obs2, reward2, *_, info2 = twin.step(act)
        text = str(info["bidding_outcomes"])                 # read at once
        kept.append((info2, text))                           # the twin's is read after its NEXT step (below)
        rows = ast.literal_eval(text)
        assert len(rows) == 24 and list(rows[0]) == ["bid", "impressions", "impression_share", "buyside_clicks", "costs", "sellside_conversions",
                                                       "revenues", "revenues_per_cost", "profit"]
        total = 0.0
        for k, r in enumerate(rows):
            assert r["bid"] == pytest.approx(float(act["keyword_bids"][k]))
            assert r["impressions"] == obs["impressions"][k] and r["buyside_clicks"] == obs["buyside_clicks"][k] == len(r["costs"])
            assert r["sellside_conversions"] == obs["sellside_conversions"][k] == len(r["revenues"])
            assert len(r["revenues_per_cost"]) == len(r["costs"]) and [x for x in r["revenues_per_cost"] if x != 0.0] == r["revenues"]
            assert 0.0 <= r["impression_share"] <= 1.0 and (r["impression_share"] > 0) == (r["impressions"] > 0)
            if implicit:     # whole cents: the lists add up to the observation exactly
                assert round(sum(r["costs"]) * 100) == round(float(obs["cost"][k]) * 100)
                assert all(abs(c * 100 - round(c * 100)) < 1e-9 for c in r["costs"])
            else:
                assert sum(r["costs"]) == pytest.approx(float(obs["cost"][k]), rel=1e-6, abs=1e-6)
            assert round(sum(r["revenues"]) * 100) == round(float(obs["revenue"][k]) * 100)
            assert r["profit"] == pytest.approx(sum(r["revenues"]) - sum(r["costs"]), abs=1e-9)
            total += r["profit"]
        assert total == pytest.approx(reward, abs=1e-6)
        spend = sum(sum(r["costs"]) for r in rows)
        assert spend <= budget + 1e-9
        bound = bound or spend > budget - 2.0
    assert bound == (budget < 1e5)                           # the small budgets really bound
    for (late_info, text_then) in kept:                      # an earlier step of the running episode replays just the same
        assert str(late_info["bidding_outcomes"]) == text_then
    env.close()
    twin.close()


def test_bidding_outcomes_cost_nothing_unread_and_expire_honestly(pkg):
    """step() does nothing for info["bidding_outcomes"]: a loop that keeps the previous info alive (the usual `obs, r, term,
    trunc, info = env.step(a)`) does not trigger a replay.  With drift on only the last step can be regenerated; after a reset
    none of the earlier ones: the late string carries the step's totals and says so."""
    import ast
    env = pkg.BiddingSimulation(keyword_config=_cfg(40, 0.5), num_keywords=16, updater_mask=[True] * 16)
    env.reset(seed=3)
    calls = []
    real = env._engine.outcomes_replay
    env._engine.outcomes_replay = lambda *a, **k: (calls.append(k.get("steps_back", 1)), real(*a, **k))[1]
    act = {"keyword_bids": np.full(16, 0.9), "budget": np.array([1.0e6])}
    infos = []
    for _ in range(3):
        obs, r, term, trunc, info = env.step(act)
        infos.append((info, obs))
    assert calls == []
    last = ast.literal_eval(str(infos[2][0]["bidding_outcomes"]))
    assert calls == [1] and "per_click" not in last[0] and len(last[0]["costs"]) == infos[2][1]["buyside_clicks"][0]
    old = ast.literal_eval(str(infos[0][0]["bidding_outcomes"]))              # drift has moved the parameters since
    assert calls == [1] and old[0]["per_click"] == "expired"
    assert [round(r["costs"][0] * 100) for r in old] == [round(float(c) * 100) for c in infos[0][1]["cost"]]
    assert [r["buyside_clicks"] for r in old] == list(infos[0][1]["buyside_clicks"])
    env.reset()
    late = ast.literal_eval(str(infos[1][0]["bidding_outcomes"]))
    assert late[0]["per_click"] == "expired" and calls == [1]
    env.close()


def test_bidding_outcomes_survive_a_keyword_params_read_under_drift(pkg):
    """With drift on the last step's update_keywords() is pending on the device until somebody needs the parameters; reading
    env.keyword_params (or str(info["keyword_params"])) writes it into the planes.  The step's per-click lists must still be the
    step's own: the facade formats a still-unread info["bidding_outcomes"] before the parameters move, and an engine asked to
    replay a step whose drift has been applied says so instead of regenerating clicks from moved parameters."""
    import ast
    env = pkg.BiddingSimulation(keyword_config=_cfg(60, 0.5), num_keywords=12, updater_mask=[True] * 12,
                                updater_params=[["vol", 0.5], ["ctr", 0.5], ["cvr", 0.5]])
    twin = pkg.BiddingSimulation(keyword_config=_cfg(60, 0.5), num_keywords=12, updater_mask=[True] * 12,
                                 updater_params=[["vol", 0.5], ["ctr", 0.5], ["cvr", 0.5]])
    env.reset(seed=11)
    twin.reset(seed=11)
    act = {"keyword_bids": np.full(12, 1.1), "budget": np.array([1.0e6])}
    for _ in range(2):
        obs, r, term, trunc, info = env.step(act)
        obs2, *_, info2 = twin.step(act)
    want = str(info2["bidding_outcomes"])                    # the twin reads its outcomes first: the step's own lists
    p_before = [list(p) for p in env._params_host]
    params = env.keyword_params                              # applies the pending drift on the device
    assert any(p[3] != q[3] for p, q in zip(params, p_before))                       # the parameters really moved
    text = str(info["bidding_outcomes"])
    assert text == want
    rows = ast.literal_eval(text)
    assert [len(r["costs"]) for r in rows] == list(obs["buyside_clicks"]) and "per_click" not in rows[0]
    # the engine's own guard (no facade in between): the drift of the last step has been applied -> ADC_ESTATE
    with pytest.raises(AssertionError, match="already been applied"):
        env._engine.outcomes_replay(0, np.full(12, 1.1, np.float32), 1.0e6)
    # a dropped info costs nothing and the text falls back honestly when nobody kept the step's own
    obs, r, term, trunc, info = env.step(act)
    env._last_outcomes = None                                # (as if the caller had only kept a copy of the dict's other entries)
    _ = env.keyword_params
    late = ast.literal_eval(str(info["bidding_outcomes"]))
    assert late[0]["per_click"] == "expired" and [r["buyside_clicks"] for r in late] == list(obs["buyside_clicks"])
    env.close()
    twin.close()


def test_outcomes_replay_refuses_after_the_stream_or_the_parameters_were_rewritten(pkg):
    """adc_engine_outcomes_replay recomputes the stream position and reads the parameters as they stand: after set_rng_state,
    set_params / set_env_params, update_keywords or a generated keyword set it returns ADC_ESTATE, not plausible lists; days run
    from a captured hipGraph count as steps."""
    from adcraft_amd.engine import StepEngine
    from tests import helpers as H
    K = 32
    planes = H.implicit_params(2, K, seed=4, mean_volume=30)
    bids = np.full((2, K), 0.8, np.float32)
    for change in ("rng", "params", "env_params", "general"):
        e = StepEngine(2, K, seed=3, model=2 if change == "general" else 0)
        e.set_all_params(planes)
        e.reset()
        e.step(bids, 1.0e6)
        assert len(e.outcomes_replay(1, bids[1], 1.0e6)["keyword"]) > 0
        if change == "rng":
            e.set_rng_state(*e.get_rng_state())
        elif change == "params":
            e.set_all_params(planes)
        elif change == "env_params":
            e.set_env_params(0, planes[:, 0])
        else:
            e.set_general_model(20, 0.5, 1)
        with pytest.raises(AssertionError):
            e.outcomes_replay(1, bids[1], 1.0e6)
        e.step(bids, 1.0e6)
        assert len(e.outcomes_replay(1, bids[1], 1.0e6)["keyword"]) > 0          # the next step is replayable again
        e.close()
    e = StepEngine(2, K, seed=3)
    e.set_all_params(planes)
    e.reset()
    e.sample_actions(0.3, 1.0, 1.0e6)
    e.run_days("fixed", 5, graph=True)                       # the same actions every day; the captured pair replays the later days
    e.synchronize()
    got_bids, got_budget = e.get_actions()
    rec = e.outcomes_replay(0, got_bids[0], float(got_budget[0]))
    out = e.fetch()
    assert np.bincount(rec["keyword"], minlength=K).tolist() == out["buyside_clicks"][0].tolist()
    e.close()


def test_reset_seed_reproduces_reference_keywords(pkg, golden):
    kat = golden("g2_keyword_params.json")["notebook_kat"]
    env = pkg.BiddingSimulation(keyword_config=_cfg(100, 0.3), num_keywords=30)
    env.reset(seed=10)
    p = env.keyword_params[0]
    assert [list(map(float, p[0]))] + [float(x) for x in p[1:]] == [[100.0, 13.0]] + kat["seed10_kw0"][1:]
    assert len(env.keywords) == 30 and env.keywords[0].buyside_ctr == p[3]
    assert env.keywords[0].sample_bids(2048).shape == (1, 2048)
    env.close()


def test_same_seed_same_trajectory_and_matches_oracle(pkg):
    from oracle import capi as orc
    K = 24
    bids = np.round(np.random.default_rng(0).uniform(0.3, 1.1, (6, K)), 2)

    def run(seed):
        env = pkg.BiddingSimulation(keyword_config=_cfg(64, 0.8), num_keywords=K, budget=40.0, max_days=6)
        env.reset(seed=seed)
        o = orc.OracleEngine(1, K, max_days=6)
        o.params[:] = env._engine.get_all_params()
        o.key[:], o.tick[:] = env._engine.get_rng_state()
        traj = []
        for t in range(6):
            obs, r, term, trunc, info = env.step({"keyword_bids": bids[t].astype(np.float32), "budget": np.float32(40.0)})
            ref = o.step(bids[t].astype(np.float32), 40.0)
            assert obs["impressions"].tolist() == ref["impressions"][0].tolist()
            assert obs["buyside_clicks"].tolist() == ref["clicks"][0].tolist()
            assert r == ref["reward"][0] and term == bool(ref["terminated"][0])
            assert abs(obs["cost"].sum() - 40.0) < 40.0 + 1e-6
            traj.append((obs["impressions"].tolist(), obs["sellside_conversions"].tolist(), r))
        assert term and env.current_day == 6
        env.close()
        return traj
    a, b, c = run(3), run(3), run(4)
    assert a == b and a != c


def test_options_truncation_render(pkg):
    env = pkg.BiddingSimulation(keyword_config=_cfg(64, 0.1), num_keywords=8, render_mode="ansi")
    env.reset(seed=2, options={"max_days": 3, "loss_threshold": 5.0})
    assert env.max_days == 3 and env.loss_threshold == 5.0
    obs, r, term, trunc, info = env.step({"keyword_bids": np.full(8, 1.5, np.float32)})
    assert trunc and r < -5.0 and env.cumulative_profit == pytest.approx(r)
    txt = env.render()
    assert txt.startswith("Time step: 1/3") and "truncated early" in txt
    assert info["bids"] == [1.5] * 8
    obs0, _ = env.reset()
    assert env.current_day == 0 and env.cumulative_profit == 0.0
    env.close()


def test_drift_keyword_params_follow_device(pkg):
    from oracle import capi as orc
    K = 12
    env = pkg.BiddingSimulation(keyword_config=_cfg(128, 0.8), num_keywords=K, updater_mask=[True] * K,
                                updater_params=[["vol", 0.05], ["ctr", 0.1], ["cvr", 0.2]])
    env.reset(seed=9)
    p0 = [list(p) for p in env.keyword_params]
    o = orc.OracleEngine(1, K, drift=(0.05, 0.1, 0.2), drift_on=True)
    o.params[:] = env._engine.get_all_params()
    o.key[:], o.tick[:] = env._engine.get_rng_state()
    for t in range(3):
        b = np.full(K, 0.8, np.float32)
        env.step({"keyword_bids": b})
        o.step(b, 1000.0)
    o.materialize_drift()
    p = env.keyword_params
    assert [q[0][0] for q in p] == [float(x) for x in o.params[orc.P_VOL_MEAN, 0]]
    assert [q[3] for q in p] == [float(x) for x in o.params[orc.P_BCTR, 0]]
    assert [q[0][1] for q in p] == [q[0][1] for q in p0] and any(q[3] != q0[3] for q, q0 in zip(p, p0))
    env.update_keywords()                       # direct call, as the reference allows
    assert any(q[3] != float(x) for q, x in zip(env.keyword_params, o.params[orc.P_BCTR, 0]))
    env.close()


def test_default_constructor_is_explicit_model_with_phantom_clicks(pkg):
    env = pkg.BiddingSimulation(num_keywords=64)        # BASELINE cfg1 shape: default config, 1 env x 64 keywords
    env.reset(seed=0)
    tot_c = tot_i = 0
    for _ in range(5):
        obs, r, term, trunc, _ = env.step({"keyword_bids": np.full(64, 0.05, np.float32)})
        tot_c += obs["buyside_clicks"].sum()
        tot_i += obs["impressions"].sum()
    assert tot_c > tot_i        # zero-impression sub-steps still yield phantom clicks (SURVEY B-1)
    env.close()


# ---------------------------------------------------------------- vector env
def test_vector_env_matches_single_envs(pkg):
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    N, K = 6, 16
    vec = BiddingSimulationVectorEnv(N, keyword_config=_cfg(64, 0.8), num_keywords=K, max_days=3)
    obs, _ = vec.reset(seed=50)
    assert obs["impressions"].shape == (N, K) and obs["cumulative_profit"].shape == (N, 1)
    bids = np.round(np.random.default_rng(1).uniform(0.3, 1.0, (N, K)), 2).astype(np.float32)
    vobs, rew, term, trunc, infos = vec.step({"keyword_bids": bids, "budget": np.full(N, 1000.0, np.float32)})
    for i in (0, 3, 5):      # env i == a single BiddingSimulation reset with seed 50 + i
        env = pkg.BiddingSimulation(keyword_config=_cfg(64, 0.8), num_keywords=K, max_days=3)
        env.reset(seed=50 + i)
        sobs, r, *_ = env.step({"keyword_bids": bids[i]})
        assert sobs["impressions"].tolist() == vobs["impressions"][i].tolist()
        assert sobs["revenue"].tolist() == vobs["revenue"][i].tolist() and r == rew[i]
        env.close()
    vec.step({"keyword_bids": bids})
    vobs, rew, term, trunc, infos = vec.step({"keyword_bids": bids})
    assert term.all() and infos["_final_obs"].all() and (infos["final_obs"]["days_passed"] == 3).all()
    # same-step autoreset: the terminal observation is in infos["final_obs"]; obs is the next episode's reset observation
    assert infos["final_obs"]["impressions"].sum() > 0
    assert all((v == 0).all() for v in vobs.values())
    vobs, *_ = vec.step({"keyword_bids": bids})
    assert (vobs["days_passed"] == 1).all()          # autoreset restarted the episodes
    vec.close()


def test_flat_layout_and_adaptors(pkg):
    from adcraft_amd import gymnasium_kw_utils as utils
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv, RLlibVectorEnvAdapter, SB3VecEnvAdapter
    N, K = 4, 8
    mk = lambda flat: BiddingSimulationVectorEnv(N, keyword_config=_cfg(64, 0.8), num_keywords=K, flat=flat, max_days=2)  # noqa: E731
    d, f = mk(False), mk(True)
    d.reset(seed=7)
    f.reset(seed=7)
    act = np.concatenate([np.full((N, 1), 500.0), np.full((N, K), 0.7)], axis=1).astype(np.float32)
    od, *_ = d.step(act)
    of, *_ = f.step(act)
    assert of.shape == (N, 5 * K + 2) and of.dtype == np.float32
    for i in range(N):
        assert np.array_equal(of[i], utils.flatten_dict_array({k: v[i] for k, v in od.items()}).astype(np.float32))
    r = RLlibVectorEnvAdapter(f)
    obs_list, infos = r.vector_reset(seeds=[7] * N)
    o2, rew, term, trunc, infos = r.vector_step([a for a in act])
    assert len(o2) == N and np.array_equal(np.stack(o2), of) and len(rew) == N
    o2b, _, term2, _, infos2 = f.step(act)                 # the flat path: same-step autoreset too (max_days = 2)
    assert term2.all() and (o2b == 0).all() and infos2["final_obs"].shape == (N, 5 * K + 2) and (infos2["final_obs"][:, 2 * K + 1] == 2).all()
    f.reset(seed=7)
    f.step(act)
    s = SB3VecEnvAdapter(mk(True))
    s.seed(7)
    o3 = s.reset()
    o3, rew3, dones, infos3 = s.step(act)
    assert np.array_equal(o3, of)
    o3, rew3, dones, infos3 = s.step(act)
    assert dones.all() and "terminal_observation" in infos3[0] and (o3 == 0).all()
    for x in (d, f):
        x.close()
    s.close()


def test_vectorised_sampler_for_large_env_counts(pkg):
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    vec = BiddingSimulationVectorEnv(2048, keyword_config=_cfg(16, 0.1, no_vol_prob=0.5), num_keywords=128,
                                     param_sampler="vectorised")
    vec.reset(seed=1)
    obs, rew, term, trunc, _ = vec.step({"keyword_bids": np.full((2048, 128), 0.8, np.float32)})
    assert obs["impressions"].shape == (2048, 128) and rew.shape == (2048,)
    frac_zero = (obs["impressions"] == 0).mean()
    assert 0.45 < frac_zero < 0.75          # half the keywords have no volume (no_vol_prob = 0.5)
    vec.close()


def test_device_keyword_generation_matches_oracle_and_reference_law(pkg):
    import ctypes as C
    from scipy import stats
    from adcraft_amd import gymnasium_kw_utils as utils
    from adcraft_amd.engine import StepEngine
    from oracle import capi as orc
    N, K = 64, 512
    table = utils.generate_simple_experiment_quantiles(16, 0.1)
    # a second bucket so that the bucket pick matters
    for name, lo, md, hi in (("vol", 100, 128, 200), ("ave_cpc", 0.2, 0.4, 0.6), ("std_cpc", 0.05, 0.1, 0.2),
                             ("bctr", 0.2, 0.3, 0.4), ("sctr", 0.5, 0.6, 0.7), ("rpsc", 1.0, 2.0, 3.0), ("std_rpsc", 0.1, 0.2, 0.3)):
        table[f"count_{name}"].append(3)
        table[f"min_{name}"].append(lo)
        table[f"median_{name}"].append(md)
        table[f"max_{name}"].append(hi)
    e = StepEngine(N, K, seed=3)
    e.reset(seeds=np.arange(N, dtype=np.uint64) + np.uint64(1000))
    e.generate_keywords(table, no_vol_prob=0.3)
    got = e.get_all_params()
    keys, _ = e.get_rng_state()
    # (1) bit-exact against the oracle's restatement of the same Philox recipe
    names = StepEngine.QUANTITIES
    arrs = [[np.ascontiguousarray(table[f"{c}_{n}"], dtype=np.float32) for n in names] for c in ("min", "median", "max")]
    ptrs = [(C.c_void_p * 7)(*[a.ctypes.data for a in arrs[i]]) for i in range(3)]
    buckets = (C.c_int32 * 7)(*[2] * 7)
    ref = np.zeros((8, N, K), dtype=np.float32)
    L = orc.lib()
    L.orc_generate_implicit_keywords.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_float, C.c_void_p]
    L.orc_generate_implicit_keywords(N, K, keys.ctypes.data, 0, buckets, ptrs[0], ptrs[1], ptrs[2], 0.3, ref.ctypes.data)
    assert np.array_equal(got, ref)
    # (2) same law as the reference's host recipe (PCG64): two-sample KS per parameter
    host = np.concatenate([utils.implicit_params_to_planes(
        utils.sample_implicit_keyword_params(K, np.random.default_rng(s), {"load_quant_func": lambda kc: table,
                                                                           "quantiles_folder": "x", "no_vol_prob": 0.3}))
        for s in range(N)], axis=1)
    for p in range(8):
        assert stats.ks_2samp(got[p].ravel(), host[p].ravel()).pvalue > 1e-4, p
    assert abs((got[0] == 0).mean() - 0.3) < 0.01
    e.close()


def test_device_explicit_keyword_generation_matches_oracle_and_reference_law(pkg):
    """f3, the other generator: sample_random_keywords (gymnasium_kw_utils.py:113-156) on the device - k_generate_explicit_keywords ==
    the oracle's restatement bit for bit (all eight planes, masked envs untouched, a second serial differs), every plane against
    numpy's beta / random at n = 2.6e5 (two-sample KS), vol_mean in B-8's range"""
    from scipy import stats
    from adcraft_amd.engine import StepEngine
    from tests.test_keygen_explicit import oracle_planes
    N, K = 512, 512
    e = StepEngine(N, K, model=1, seed=3)
    e.reset(seeds=np.arange(N, dtype=np.uint64) + np.uint64(4000))
    e.generate_explicit_keywords()
    got = e.get_all_params()
    keys, _ = e.get_rng_state()
    assert np.array_equal(got, oracle_planes(keys, K, 0))
    mask = np.zeros(N, np.uint8)
    mask[::3] = 1
    e.generate_explicit_keywords(env_mask=mask, serial=1)
    again = e.get_all_params()
    ref1 = oracle_planes(keys, K, 1)
    assert np.array_equal(again[:, mask == 1], ref1[:, mask == 1]) and np.array_equal(again[:, mask == 0], got[:, mask == 0])
    assert not np.array_equal(ref1[:, mask == 1], got[:, mask == 1])
    host = H.explicit_params(N, K, seed=99).reshape(8, -1)
    for p in range(8):
        assert stats.ks_2samp(got[p].ravel(), host[p]).pvalue > 1e-3, p
    assert got[0].min() >= 14 and got[0].max() <= 29
    # the engine steps on them (the default-constructor model) and equals the oracle on the generated planes
    o = H.mirror_oracle(e, again)
    bids = o.sample_bids(0.3, 1.0)
    H.assert_step_equal(e.step(bids, 1.0e9), o.step(bids, 1.0e9), implicit=False)
    e.close()
    # an IMPLICIT engine refuses (its keywords come from quantile tables)
    e = StepEngine(2, 8, model=0, seed=1)
    e.reset(seeds=np.arange(2, dtype=np.uint64))
    with pytest.raises((ValueError, AssertionError)):
        e.generate_explicit_keywords()
    e.close()


def test_default_constructor_vector_env_resets_on_the_device(pkg):
    """a 4096-env default-constructor BiddingSimulationVectorEnv(param_sampler="device") resets without a host loop; the
    vectorised host sampler draws the same law"""
    from scipy import stats
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    vec = BiddingSimulationVectorEnv(4096, num_keywords=64, param_sampler="device")
    vec.reset(seed=5)
    p1 = vec.engine.get_all_params()
    assert p1[0].min() >= 14 and p1[0].max() <= 29 and 0.0 <= p1[4].min() and p1[5].max() <= 1.0
    obs, rew, term, trunc, _ = vec.step({"keyword_bids": np.full((4096, 64), 0.8, np.float32)})
    assert obs["impressions"].shape == (4096, 64) and obs["impressions"].sum() > 0 and rew.shape == (4096,)
    vec.reset(seed=5)
    assert np.array_equal(vec.engine.get_all_params(), p1)          # same seed, same keyword sets
    vec.reset(seed=6)
    assert not np.array_equal(vec.engine.get_all_params(), p1)
    vec.close()
    host = BiddingSimulationVectorEnv(4096, num_keywords=64, param_sampler="vectorised")
    host.reset(seed=5)
    p2 = host.engine.get_all_params()
    for p in range(8):
        assert stats.ks_2samp(p1[p].ravel(), p2[p].ravel()).pvalue > 1e-3, p
    host.close()


def test_vector_env_device_sampler(pkg):
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    vec = BiddingSimulationVectorEnv(4096, keyword_config=_cfg(128, 0.8), num_keywords=64, param_sampler="device")
    vec.reset(seed=9)
    p1 = vec.engine.get_all_params()
    assert (p1[0] == 128).all() and (p1[5] == np.float32(0.8)).all() and p1[2].min() >= 0.3 and p1[2].max() <= 1.0
    obs, rew, term, trunc, _ = vec.step({"keyword_bids": np.full((4096, 64), 0.8, np.float32)})
    assert obs["impressions"].mean() > 20
    vec.reset(seed=9)
    assert np.array_equal(vec.engine.get_all_params(), p1)          # same seed, same keyword sets
    vec.close()


def test_vector_env_split_over_several_engines_gives_the_same_steps(pkg):
    """engine_shards: the envs held by several engines on the device and stepped together (asynchronous host steps from
    page-locked buffers, transfers overlapping kernels) - observation for observation what one engine returns"""
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    N, K = 10, 24
    rng = np.random.default_rng(3)
    outs = []
    for shards, flat in [(1, False), (3, False), (1, True), (4, True)]:
        vec = BiddingSimulationVectorEnv(N, keyword_config=_cfg(48, 0.6), num_keywords=K, max_days=3, engine_shards=shards, flat=flat,
                                         param_sampler="vectorised")
        vec.reset(seed=11)
        rng = np.random.default_rng(3)
        traj = []
        for _ in range(5):                                    # crosses an episode end (autoreset) at day 3
            bids = np.round(rng.uniform(0.3, 1.0, (N, K)), 2).astype(np.float32)
            act = np.concatenate([np.full((N, 1), 40.0, np.float32), bids], axis=1) if flat else {"keyword_bids": bids, "budget": np.full(N, 40.0, np.float32)}
            obs, rew, term, trunc, infos = vec.step(act)
            traj.append((obs.copy() if flat else {k: v.copy() for k, v in obs.items()}, rew.copy(), term.copy(), trunc.copy()))
        outs.append(traj)
        vec.close()
    for a, b in ((outs[0], outs[1]), (outs[2], outs[3])):
        for (oa, ra, ta, ua), (ob, rb, tb, ub) in zip(a, b):
            if isinstance(oa, dict):
                assert all(np.array_equal(oa[k], ob[k]) for k in oa)
            else:
                assert np.array_equal(oa, ob)
            assert np.array_equal(ra, rb) and np.array_equal(ta, tb) and np.array_equal(ua, ub)
    assert outs[0][2][2].all() and outs[0][0][0]["impressions"].sum() > 0


def test_compact_counts_are_the_same_counts_as_uint16(pkg):
    """compact_counts: the three count observations packed to uint16 on the device (adc_step_out.counts_u16) - the same
    numbers, and a count beyond 65535 raises instead of being clipped silently"""
    from adcraft_amd.engine import MODEL_IMPLICIT, P_VOL_MEAN, StepEngine
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    N, K = 10, 24
    trajs = []
    for compact, shards in [(False, 1), (True, 1), (True, 3)]:
        vec = BiddingSimulationVectorEnv(N, keyword_config=_cfg(48, 0.6), num_keywords=K, max_days=3, engine_shards=shards,
                                         compact_counts=compact, param_sampler="vectorised")
        vec.reset(seed=5)
        rng = np.random.default_rng(8)
        traj = []
        for _ in range(4):
            bids = np.round(rng.uniform(0.3, 1.0, (N, K)), 2).astype(np.float32)
            act = {"keyword_bids": bids, "budget": np.full(N, 40.0, np.float32)}
            if shards == 3:                         # actions written in place into the engine's page-locked buffers
                buf = vec.action_buffers()
                buf["keyword_bids"][...], buf["budget"][...] = act["keyword_bids"], act["budget"]
                act = buf
            obs, rew, term, trunc, _ = vec.step(act)
            traj.append(({k: np.array(v) for k, v in obs.items()}, rew.copy()))
        trajs.append(traj)
        vec.close()
    assert trajs[1][0][0]["impressions"].dtype == np.uint16 and trajs[0][0][0]["impressions"].sum() > 0
    for other in trajs[1:]:
        for (oa, ra), (ob, rb) in zip(trajs[0], other):
            assert all(np.array_equal(oa[k], ob[k]) for k in oa) and np.array_equal(ra, rb)
    eng = StepEngine(2, 4, MODEL_IMPLICIT, compact_counts=True, seed=1)
    planes = np.zeros((8, 2, 4), np.float32)
    planes[P_VOL_MEAN] = 70000.0
    planes[2], planes[3], planes[4], planes[5], planes[6] = 0.0, 0.1, 0.5, 0.5, 1.0
    eng.set_all_params(planes)
    eng.reset()
    with pytest.raises(OverflowError):
        eng.step(np.full((2, 4), 5.0, np.float32), np.full(2, 1e9, np.float32))
    eng.close()


def test_profiling_can_sample_and_sharded_engine_names_what_it_lacks(pkg):
    """adc_engine_profile_sample_every: events around every n-th step only (measured launches = ceil(steps / n), same results);
    a ShardedStepEngine refuses the single-engine calls by name instead of failing on a missing attribute"""
    from adcraft_amd.engine import ShardedStepEngine, StepEngine
    N, K = 6, 32
    planes = np.broadcast_to(np.array([40, 6, 0.5, 0.2, 0.3, 0.5, 1.0, 0.2], np.float32)[:, None, None], (8, N, K)).copy()
    outs = []
    for every in (1, 3):
        e = StepEngine(N, K, seed=5)
        e.set_all_params(planes)
        e.reset()
        e.sample_actions(0.3, 1.0, 1e9)
        e.profile_enable(True, every=every)
        e.profile_read()
        for _ in range(7):
            e.step_device()
        ms, launches = e.profile_read()
        assert launches == (7 if every == 1 else 3) and ms[0] > 0.0
        outs.append(e.fetch())
        e.close()
    assert all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
    with pytest.raises(ValueError):
        StepEngine(1, 4).profile_enable(True, every=0)
    s = ShardedStepEngine(N, K, shards=2, seed=5)
    with pytest.raises(NotImplementedError, match="engine_shards=1"):
        s.agent_init(1.0, None)
    with pytest.raises(AttributeError):
        s.no_such_call
    s.close()
