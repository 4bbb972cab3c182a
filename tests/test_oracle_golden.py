"""The CPU oracle against the golden vectors recorded from the reference (tools/gen_golden.py).

This is what pins the oracle: integers bit-exact, money exact in cents, f64 sums within 1e-9.
"""
import numpy as np
import pytest

from oracle import capi as orc
from oracle import ref_numpy as rn


def _rng(seed):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


# ---------------------------------------------------------------- G1 nth_price_auction
def test_g1_nth_price_auction(golden):
    g = golden("g1_nth_price_auction.json")
    assert len(g["cases"]) >= 20
    for c in g["cases"]:
        ob = np.array(c["other_bids"], dtype=np.float64).reshape(-1, max(1, len(c["other_bids"][0]) if c["other_bids"] else 1))
        imp, pl, co = orc.nth_price_auction(c["bid"], ob, c["n"], c["num_winners"])
        assert imp == c["impressions"], c["tag"]
        assert pl.tolist() == c["placements"], c["tag"]
        assert co.tolist() == c["costs"], c["tag"]          # bit-exact f64


# ---------------------------------------------------------------- G2 keyword parameter generation
def test_g2_implicit_params_bit_exact(golden):
    g = golden("g2_keyword_params.json")
    for c in g["cases"]:
        params = rn.sample_implicit_params(c["K"], _rng(c["seed"]), c["quantile_table"],
                                           0.0 if c["no_vol_prob"] is None else c["no_vol_prob"])
        got = [[[float(p[0][0]), float(p[0][1])]] + [float(x) for x in rn.printed_implicit_tuple(p)[1:]] for p in params]
        assert got == c["params"]


def test_g2_quantile_table_matches_reference(golden):
    g = golden("g2_keyword_params.json")
    c = g["cases"][0]
    t = rn.simple_experiment_quantiles(c["mean_volume"], c["conversion_rate"])
    assert {k: [float(x) for x in v] for k, v in t.items()} == {k: [float(x) for x in v] for k, v in c["quantile_table"].items()}


def test_g2_notebook_known_answers(golden):
    """Parameter tuples printed in the reference's notebooks."""
    kat = golden("g2_keyword_params.json")["notebook_kat"]
    p = rn.sample_implicit_params(30, _rng(10), rn.simple_experiment_quantiles(100, 0.3))[0]
    t = rn.printed_implicit_tuple(p)
    assert [list(map(float, t[0]))] + [float(x) for x in t[1:]] == [[100.0, 13.0]] + kat["seed10_kw0"][1:]
    p = rn.sample_implicit_params(2, _rng(0), rn.simple_experiment_quantiles(16, 0.5))[0]
    t = rn.printed_implicit_tuple(p)
    assert [list(map(float, t[0]))] + [float(x) for x in t[1:]] == [[16.0, 1.0]] + kat["seed0_kw0"][1:]


def test_g2_explicit_params_bit_exact(golden):
    for c in golden("g2_explicit_params.json")["cases"]:
        params = rn.sample_random_params(c["K"], _rng(c["seed"]))
        got = [[[float(p[0][0]), float(p[0][1])]] + [float(x) for x in p[1:]] for p in params]
        assert got == c["params"]


# ---------------------------------------------------------------- G3 campaign replay (IMPLICIT)
def _load_implicit(eng, kp):
    for k, p in enumerate(kp):
        eng.params[orc.P_VOL_MEAN, 0, k] = p["vol_mean"]
        eng.params[orc.P_VOL_STD, 0, k] = p["vol_std"]
        eng.params[orc.P_A, 0, k] = p["loc"]
        eng.params[orc.P_B, 0, k] = p["scale"]
        eng.params[orc.P_BCTR, 0, k] = p["bctr"]
        eng.params[orc.P_SCTR, 0, k] = p["sctr"]
        eng.params[orc.P_REV_MEAN, 0, k] = p["rev_mean"]
        eng.params[orc.P_REV_STD, 0, k] = p["rev_std"]


def test_g3_implicit_replay_bit_exact(golden):
    g = golden("g3_implicit_replay.json")
    assert len(g["traces"]) >= 9
    n_binding = 0
    for t in g["traces"]:
        K = t["K"]
        eng = orc.OracleEngine(1, K, model=orc.IMPLICIT)
        _load_implicit(eng, t["keyword_params"])
        tape = orc.TapeSource(bid_cents=t["tape"]["bid"], click=t["tape"]["click"], conv=t["tape"]["conv"],
                              rev_cents=t["tape"]["rev"])
        tape.set_volumes(np.array(t["volumes"]).reshape(1, K))
        o = eng.step(np.array(t["bids"], dtype=np.float32), t["budget"], tape)
        ref = t["out"]
        assert o["impressions"][0].tolist() == ref["impressions"]
        assert o["clicks"][0].tolist() == ref["buyside_clicks"]
        assert o["conversions"][0].tolist() == ref["sellside_conversions"]
        np.testing.assert_allclose(o["cost"][0], ref["cost"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(o["revenue"][0], ref["revenue"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(o["reward"][0], sum(ref["profit"]), rtol=0, atol=1e-9)
        # the oracle consumed exactly the variates the reference drew
        cur = tape.cursors()
        assert cur["bid"] == len(t["tape"]["bid"]) and cur["click"] == len(t["tape"]["click"])
        assert cur["conv"] == len(t["tape"]["conv"]) and cur["rev"] == len(t["tape"]["rev"])
        n_binding += t["budget"] < 1e8
    assert n_binding >= 5


def test_g3_explicit_replay(golden):
    g = golden("g3_explicit_replay.json")
    for t in g["traces"]:
        K = t["K"]
        eng = orc.OracleEngine(1, K, model=orc.EXPLICIT)
        tape = orc.TapeSource(click=t["tape"]["click"], conv=t["tape"]["conv"], rev_cents=t["tape"]["rev"],
                              x_impressions=t["tape"]["impressions"], x_cost=t["tape"]["cost"])
        tape.set_volumes(np.array(t["volumes"]).reshape(1, K))
        o = eng.step(np.array(t["bids"], dtype=np.float32), t["budget"], tape)
        ref = t["out"]
        assert o["impressions"][0].tolist() == ref["impressions"]
        assert o["clicks"][0].tolist() == ref["buyside_clicks"]          # includes the phantom clicks (B-1)
        assert o["conversions"][0].tolist() == ref["sellside_conversions"]
        assert o["cost"][0].tolist() == ref["cost"]                       # same f64 operation order: bit-exact
        np.testing.assert_allclose(o["revenue"][0], ref["revenue"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(o["reward"][0], sum(ref["profit"]), rtol=0, atol=1e-9)
        cur = tape.cursors()
        assert cur["ximp"] == len(t["tape"]["impressions"]) and cur["xcost"] == len(t["tape"]["cost"])
        assert cur["click"] == len(t["tape"]["click"]) and cur["conv"] == len(t["tape"]["conv"])
    # the phantom-click quirk is really exercised
    assert any(c > i for t in g["traces"] for c, i in zip(t["out"]["buyside_clicks"], t["out"]["impressions"]))


# ---------------------------------------------------------------- the per-click lists of info["bidding_outcomes"]
def test_g3_outcome_lists_element_for_element(golden):
    """f4: the combined BiddingOutcomes the reference builds (costs, revenues, revenues_per_cost per click in its order;
    impression_share with combine_outcomes' lossy denominator; profit added up sub-timestep by sub-timestep -
    bidding_simulation.py:10-38,97-147) replayed by the oracle from the reference's tapes: every list element for element"""
    from tests import helpers as H
    n_clicks = n_lossy = 0
    for t in golden("g3_implicit_replay.json")["traces"]:
        K = t["K"]
        eng = orc.OracleEngine(1, K, model=orc.IMPLICIT)
        _load_implicit(eng, t["keyword_params"])
        tape = orc.TapeSource(bid_cents=t["tape"]["bid"], click=t["tape"]["click"], conv=t["tape"]["conv"], rev_cents=t["tape"]["rev"])
        tape.set_volumes(np.array(t["volumes"]).reshape(1, K))
        o, lists = eng.step_outcomes(np.array(t["bids"], dtype=np.float32), t["budget"], tape, capacity=64)   # (64: the regrow path too)
        H.assert_outcome_lists(lists, t["out"], K)
        assert [len(c) for c in lists["costs"][0]] == t["out"]["buyside_clicks"] == o["clicks"][0].tolist()
        n_clicks += sum(t["out"]["buyside_clicks"])
        # B-11: a sub-timestep without impressions drops out of the denominator - the share is NOT impressions / volume then
        n_lossy += sum(1 for k in range(K) if t["volumes"][k] > 0 and t["out"]["impressions"][k] > 0
                       and t["out"]["impression_share"][k] != t["out"]["impressions"][k] / t["volumes"][k])
    assert n_clicks > 1000 and n_lossy > 10
    n_phantom = 0
    for t in golden("g3_explicit_replay.json")["traces"]:
        K = t["K"]
        eng = orc.OracleEngine(1, K, model=orc.EXPLICIT)
        tape = orc.TapeSource(click=t["tape"]["click"], conv=t["tape"]["conv"], rev_cents=t["tape"]["rev"],
                              x_impressions=t["tape"]["impressions"], x_cost=t["tape"]["cost"])
        tape.set_volumes(np.array(t["volumes"]).reshape(1, K))
        o, lists = eng.step_outcomes(np.array(t["bids"], dtype=np.float32), t["budget"], tape)
        H.assert_outcome_lists(lists, t["out"], K)
        n_phantom += sum(c.count(0.0) for c in t["out"]["costs"])
    assert n_phantom > 0                      # the zero-cost phantom click (B-1) is listed, as the reference lists it


def test_g8_outcome_lists_every_step(golden):
    """the same for whole step() episodes: what BiddingSimulation.step hands to rust.repr_outcomes_py (gymnasium_kw_env.py:249)"""
    from tests import helpers as H
    for ep in golden("g8_env_episodes.json")["episodes"]:
        K = ep["K"]
        eng = orc.OracleEngine(1, K, model=orc.IMPLICIT, max_days=ep["max_days"], loss_threshold=ep["loss_threshold"])
        tp = ep["tape"]
        tape = orc.TapeSource(bid_cents=tp["bid"], click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"])
        params = ep["params0"]
        for st in ep["steps"]:
            eng.params[:8, 0] = np.array([[p[0][0] for p in params], [p[0][1] for p in params], [p[1] for p in params],
                                          [np.float32(1.0 / p[2]) for p in params], [p[3] for p in params], [p[4] for p in params],
                                          [p[5] for p in params], [p[6] for p in params]], np.float32)
            tape.set_volumes(np.array(st["volumes"]).reshape(1, K))
            o, lists = eng.step_outcomes(np.array(st["bids"], dtype=np.float32), st["budget"], tape)
            H.assert_outcome_lists(lists, st["outcomes"], K)
            assert o["impressions"][0].tolist() == st["obs"]["impressions"]
            params = st["params_after"]


# ---------------------------------------------------------------- G12: the reference's default ImplicitKeyword
def test_g12_general_implicit_replay(golden):
    """the non-env ImplicitKeyword (B ~ Binomial bidders per call, raw Laplace bids, literal top-(w+n) second-price clearing:
    synthetic_kw_classes.py:610-686, synthetic_kw_helpers.py:116-180) in the reference's campaign loop, replayed from the
    variates the reference drew: integers exact, float64 cost sums bit-exact (same operation order), cursors at the tape ends"""
    g = golden("g12_implicit_general_replay.json")
    for t in g["traces"]:
        K = t["K"]
        eng = orc.OracleEngine(1, K, model=orc.IMPLICIT_GENERAL, max_bidders=t["max_bidders"], participation_rate=t["participation_rate"])
        kp = t["keyword_params"]
        eng.params[2, 0] = [p["bid_loc"] for p in kp]
        eng.params[3, 0] = [p["bid_scale"] for p in kp]
        eng.params[4, 0] = [p["bctr"] for p in kp]
        eng.params[5, 0] = [p["sctr"] for p in kp]
        eng.params[6, 0] = [p["rev_mean"] for p in kp]
        eng.params[7, 0] = [p["rev_std"] for p in kp]
        tp = t["tape"]
        tape = orc.TapeSource(click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"], x_impressions=tp["bidders"], x_cost=tp["bids"])
        tape.set_volumes(np.array(t["volumes"]).reshape(1, K))
        o = eng.step(np.array(t["bids"], dtype=np.float32), t["budget"], tape)
        ref = t["out"]
        assert o["impressions"][0].tolist() == ref["impressions"]
        assert o["clicks"][0].tolist() == ref["buyside_clicks"]
        assert o["conversions"][0].tolist() == ref["sellside_conversions"]
        assert o["cost"][0].tolist() == ref["cost"]                       # same f64 operation order: bit-exact
        np.testing.assert_allclose(o["revenue"][0], ref["revenue"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(o["reward"][0], sum(ref["profit"]), rtol=0, atol=1e-9)
        cur = tape.cursors()
        assert cur["ximp"] == len(tp["bidders"]) and cur["xcost"] == len(tp["bids"])
        assert cur["click"] == len(tp["click"]) and cur["conv"] == len(tp["conv"]) and cur["rev"] == len(tp["rev"])
    assert any(min(t["tape"]["bidders"]) < 3 for t in g["traces"])       # the zero-padding case (fewer bidders than w+n) occurs
    assert any(t["budget"] < 1e8 and sum(t["out"]["buyside_clicks"]) > 0 for t in g["traces"])


def _g12_oracle(t):
    K = t["K"]
    eng = orc.OracleEngine(1, K, model=orc.IMPLICIT_GENERAL, max_bidders=t["max_bidders"], participation_rate=t["participation_rate"])
    kp = t["keyword_params"]
    for plane, name in ((2, "bid_loc"), (3, "bid_scale"), (4, "bctr"), (5, "sctr"), (6, "rev_mean"), (7, "rev_std")):
        eng.params[plane, 0] = [p[name] for p in kp]
    tp = t["tape"]
    tape = orc.TapeSource(click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"], x_impressions=tp["bidders"], x_cost=tp["bids"])
    tape.set_volumes(np.array(t["volumes"]).reshape(1, K))
    return eng, tape


def test_g12_outcome_lists_element_for_element(golden):
    """the default ImplicitKeyword's combined BiddingOutcomes (bidding_simulation.py:10-38,124-147) as the reference recorded them
    (tools/gen_golden_general.py): every click's float64 price and revenue in order, impression_share with the lossy denominator"""
    from tests import helpers as H
    n = 0
    for t in golden("g12_implicit_general_replay.json")["traces"]:
        eng, tape = _g12_oracle(t)
        o, lists = eng.step_outcomes(np.array(t["bids"], dtype=np.float32), t["budget"], tape, capacity=32)
        H.assert_outcome_lists(lists, t["out"], t["K"])
        assert [len(c) for c in lists["costs"][0]] == t["out"]["buyside_clicks"]
        n += sum(t["out"]["buyside_clicks"])
    assert n > 1000


def test_general_implicit_stream_law():
    """the engine's own stream for that model: bidders per call ~ Binomial(30, 0.6), the winning price is the highest raw
    Laplace(0, 0.1) competitor bid - checked on the win rate and the mean price at a given bid against a numpy simulation of
    the reference's samplers (rng.binomial, rng.laplace; synthetic_kw_classes.py:659-665, 681-686)"""
    N, K, V = 64, 8, 240
    eng = orc.OracleEngine(N, K, model=orc.IMPLICIT_GENERAL, threads=4)
    eng.params[0] = V
    eng.params[1] = 0.0
    eng.params[2] = 0.0
    eng.params[3] = 0.1
    eng.params[4] = 1.0            # every impression clicked
    eng.params[5] = 0.0
    eng.params[6] = 1.0
    eng.params[7] = 0.1
    eng.key[:] = np.arange(1, N + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    bid = 0.30
    o = eng.step(np.full((N, K), bid, np.float32), 1e9)
    n = N * K * V
    win_rate = o["impressions"].sum() / n
    price = o["cost"].sum() / max(o["clicks"].sum(), 1)
    rng = np.random.default_rng(7)
    wins, costs = 0, 0.0
    for _ in range(2000):                      # 2000 calls of 24 x 10 auctions, B drawn per call as the reference does
        B = rng.binomial(30, 0.6)
        m = rng.laplace(0.0, 0.1, (B, 240)).max(axis=0) if B > 0 else np.zeros(240)
        if B < 3:
            m = np.maximum(m, 0.0)
        w = bid > m
        wins += int(w.sum())
        costs += float(m[w].sum())
    ref_rate, ref_price = wins / (2000 * 240), costs / max(wins, 1)
    assert abs(win_rate - ref_rate) < 0.01 and abs(price - ref_price) < 0.004


# ---------------------------------------------------------------- G4 drift (f64 restatement)
def test_g4_update_keywords(golden):
    for s in golden("g4_update_keywords.json")["sequences"]:
        params = [[tuple(p[0])] + p[1:] for p in s["params0"]]
        init = None
        for st in s["steps"]:
            params, init = rn.update_keywords(params, st["uniforms"], init)
            got = [[[float(p[0][0]), float(p[0][1])]] + [float(x) for x in p[1:]] for p in params]
            assert got == st["params"]


def test_g4_update_keywords_through_the_c_oracle_tape(golden):
    """G4 replayed by oracle/adcraft_oracle.c itself (not only the f64 numpy restatement): the tape carries the three
    uniform vectors, the replayed zero-volume day ends with update_keywords() on them.  The C oracle (like the HIP engine)
    holds parameters in float32: one f32 fma / multiply + clip per update, so the difference from the reference's float64
    after s steps is bounded by about s * 2^-23 relative.  Tolerance: rtol 2e-6 (+ atol 2e-6 on the volume)."""
    for seq in golden("g4_update_keywords.json")["sequences"]:
        K = seq["K"]
        up = dict((n, v) for n, v in seq["updater_params"])
        o = orc.OracleEngine(1, K, drift=(up["vol"], up["ctr"], up["cvr"]), drift_on=True)
        p0 = seq["params0"]
        o.params[:] = np.array([[p[0][0] for p in p0], [p[0][1] for p in p0], [p[1] for p in p0], [1.0 / p[2] for p in p0],
                                [p[3] for p in p0], [p[4] for p in p0], [p[5] for p in p0], [p[6] for p in p0]], np.float32).reshape(8, 1, K)
        std0 = o.params[1].copy()
        for st in seq["steps"]:
            ts = orc.TapeSource()
            ts.set_volumes(np.zeros((1, K), np.int32))
            ts.set_drift_uniforms(np.array(st["uniforms"]).reshape(3, 1, K))
            out = o.step(np.full((1, K), 0.5, np.float32), 10.0, tape=ts)
            assert out["impressions"].sum() == 0 and not o.drift_pending.any()
            ref = st["params"]
            np.testing.assert_allclose(o.params[0, 0], [p[0][0] for p in ref], rtol=2e-6, atol=2e-6)
            assert np.array_equal(o.params[1], std0)
            np.testing.assert_allclose(o.params[4, 0], [p[3] for p in ref], rtol=2e-6)
            np.testing.assert_allclose(o.params[5, 0], [p[4] for p in ref], rtol=2e-6)
            np.testing.assert_allclose(o.params[4, 0], st["kw_bctr"], rtol=2e-6)


def test_g4_uniform_draws_are_the_env_generator(golden):
    """reset(seed) then update_keywords(): the three uniform vectors come from np_random in order vol, ctr, cvr."""
    s = golden("g4_update_keywords.json")["sequences"][0]
    rng = _rng(s["seed"])
    rn.sample_implicit_params(s["K"], rng, rn.simple_experiment_quantiles(s["mean_volume"], s["conversion_rate"]))
    # keyword construction validates each reward sampler with rds(2), rds(5), rds(5)
    # (synthetic_kw_classes.py:337-339): 12 normal draws per keyword before the first step
    for _ in range(s["K"]):
        rng.normal(0, 1, 2), rng.normal(0, 1, 5), rng.normal(0, 1, 5)
    draws = [rng.uniform(-v[1], v[1], size=(s["K"],)).tolist() for v in s["updater_params"]]
    assert draws == s["steps"][0]["uniforms"]


# ---------------------------------------------------------------- G5 metrics
def test_g5_metrics(golden):
    g = golden("g5_metrics.json")
    for c in g["akncp_ncp"]:
        kp, ip = np.array(c["kw_profits"]), np.array(c["ideal_profits"])
        assert float(rn.compute_AKNCP(kp, ip)) == c["AKNCP"]
        assert float(rn.compute_NCP(kp, ip)) == c["NCP"]
    for c in g["max_expected"]:
        r = rn.max_expected_bid_profits(c["kw_params"], np.array(c["cpc"]), np.array(c["ir"]))
        assert (float(r[0]), float(r[1]), r[2]) == (c["max_profit"], c["frac_positive"], c["argmax"])
    bids = np.array(g["bid_array"])
    for c in g["bid_curves"]:
        samples = (np.array(c["samples_cents"], dtype=np.float64) / 100.0).reshape(1, -1)
        ir, cpc = rn.implicit_bid_cpc_impressions(samples, bids)
        assert ir.tolist() == c["impression_rates"]
        np.testing.assert_allclose(cpc, c["cpc"], rtol=1e-12)


# ---------------------------------------------------------------- G6 / G7
def test_g6_flatten(golden):
    g = golden("g6_flatten.json")
    obs = {k: np.array(v) for k, v in g["obs"].items()}
    assert rn.flatten_dict_array(obs).tolist() == g["flat"]
    assert sorted(obs) == g["key_order"]


def test_g7_kat_tables(golden):
    g = golden("g7_kat_tables.json")
    for x, s, t, exp in g["sigmoid"]["rows"]:
        assert round(float(rn.sigmoid(x, s, t)), 4) == exp
        assert round(orc.lib().orc_sigmoid_f64(x, s, t), 4) == exp
    for x, exp in g["probify"]["rows"]:
        assert rn.probify(x) == exp
    assert rn.probify(np.array(g["probify"]["array"][0])).tolist() == g["probify"]["array"][1]
    for x, exp in g["nonnegify"]["rows"]:
        assert rn.nonnegify(x) == exp
    assert rn.nonnegify(np.array(g["nonnegify"]["array"][0])).tolist() == g["nonnegify"]["array"][1]
    for x, exp in g["beta_param"]["rows"]:
        assert rn.beta_param(x) == exp
    with pytest.raises(ZeroDivisionError):
        rn.beta_param(0)
    for x, exp in g["sum_list"]["rows"]:
        assert rn.sum_list(x) == exp
    s = g["seeded_samplers"]
    rng = _rng(77)
    assert rn.bid_abs_laplace(0.55, 0.08, rng)(1, 32).tolist() == s["bid_abs_laplace"]["out"]
    assert rn.rev_normal(1.0, 0.15, rng)(32).tolist() == s["rev_normal"]["out"]
    assert [int(x) for x in rn.coinflips(0.3, 32, rng)] == s["coinflips"]["out"]
    assert rn.bid_abs_normal(0.4, 0.2, _rng(78), 0.05)(2, 8).tolist() == s["bid_abs_normal"]["out"]


# ---------------------------------------------------------------- G8 whole step() episodes
def test_g8_env_episodes(golden):
    for ep in golden("g8_env_episodes.json")["episodes"]:
        K = ep["K"]
        eng = orc.OracleEngine(1, K, model=orc.IMPLICIT, max_days=ep["max_days"], loss_threshold=ep["loss_threshold"])
        tp = ep["tape"]
        tape = orc.TapeSource(bid_cents=tp["bid"], click=tp["click"], conv=tp["conv"], rev_cents=tp["rev"])
        for i, st in enumerate(ep["steps"]):
            tape.set_volumes(np.array(st["volumes"]).reshape(1, K))
            o = eng.step(np.array(st["bids"], dtype=np.float32), st["budget"], tape)
            ob = st["obs"]
            assert o["impressions"][0].tolist() == ob["impressions"]
            assert o["clicks"][0].tolist() == ob["buyside_clicks"]
            assert o["conversions"][0].tolist() == ob["sellside_conversions"]
            np.testing.assert_allclose(o["cost"][0], ob["cost"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(o["revenue"][0], ob["revenue"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(o["reward"][0], st["reward"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(o["cum_profit"][0], ob["cumulative_profit"][0], rtol=0, atol=1e-9)
            assert o["day"][0] == ob["days_passed"][0] == i + 1
            assert bool(o["terminated"][0]) == st["terminated"] and bool(o["truncated"][0]) == st["truncated"]
            cur = tape.cursors()
            sl = st["tape_slices"]
            assert (cur["bid"], cur["click"], cur["conv"], cur["rev"]) == (sl["bid"][1], sl["click"][1], sl["conv"][1], sl["rev"][1])
    assert any(st["truncated"] for ep in golden("g8_env_episodes.json")["episodes"] for st in ep["steps"])


def test_g10_zero_margin_agent_restatement(golden):
    """G10: the reference's NaiveZeroMarginStrategy, run unmodified in the notebooks' loop; the restatement must
    return the same float64 bids, the same budget and hold the same caches after every step."""
    for c in golden("g10_zero_margin_agent.json")["cases"]:
        a = rn.ZeroMarginAgent(1, c["K"], c["default_rpc"])
        for t, s in enumerate(c["steps"]):
            a.update(np.array([s["obs_clicks"]]), np.array([s["obs_conversions"]]), np.array([s["obs_revenue"]]))
            u = np.array([s["uniforms"]])
            assert np.array_equal(np.isfinite(u[0]), a.num_rpc_obs[0] < 1)        # it draws exactly where no rpc was seen
            bids, budget = a.act(np.where(np.isnan(u), 0.5, u))
            assert np.array_equal(bids[0], np.array(s["bids"])), t
            assert budget[0] == s["budget"]
            assert np.array_equal(a.ave_rpc[0].astype(np.float64), np.array(s["ave_rpc"]))
            assert np.array_equal(a.num_rpc_obs[0], np.array(s["num_rpc_obs"]))
            seen = a.num_sctr_obs[0] > 0
            assert np.array_equal(a.ave_sctr[0].astype(np.float64)[seen], np.array(s["ave_sctr"])[seen])
            assert np.array_equal(a.num_sctr_obs[0].astype(np.float64), np.array(s["num_sctr_obs"]))
            assert np.array_equal(a.max_bids[0], np.array(s["max_bids"]))
