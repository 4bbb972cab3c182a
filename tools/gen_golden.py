#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (it needs /root/reference); nothing here is
imported by the product, the oracle, the tests or the bench.  The reference's
pure-Python hot path (synthetic_kw_helpers / synthetic_kw_classes /
bidding_simulation / gymnasium_kw_utils / gymnasium_kw_env / experiment_metrics)
is imported UNMODIFIED from /root/reference.  Two of its imports do not exist in
this image, so they are stood in *in memory* (objects in sys.modules, no files):

  * ``adcraft.rust`` (pyo3 extension, un-buildable: no cargo/rustc) - only the
    four reducers whose semantics the reference's own tests pin
    (adcraft/tests/rust/test_numpy_funcs.py:10-132) plus inert placeholders that
    are never allowed to influence a recorded number: every Rust sampler on the
    path is *replaced by a recorded, injected sampler* (KeywordParams lets the
    caller inject ``volume_sampler`` / ``cost_per_buyside_click``,
    adcraft/synthetic_kw_classes.py:90-117).
  * ``gymnasium`` (not installed) - ``Env`` with the seeding contract
    ``np_random = Generator(PCG64(SeedSequence(seed)))`` and ``Box``/``Dict``
    shells.

What is written is data only: inputs, the variates the reference actually drew
("tapes"), and the outputs it computed from them.

Usage:  python tools/gen_golden.py            (rewrites tests/golden/*.json)
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# --------------------------------------------------------------------------- stand-ins
def install_standins():
    rust = types.ModuleType("adcraft.rust")

    def sum_array_bool(x):
        if not (isinstance(x, np.ndarray) and x.dtype == np.bool_):
            raise TypeError("expected ndarray[bool]")
        return int(np.count_nonzero(x))

    def sum_array(x):
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64):
            raise TypeError("expected ndarray[f64]")
        s = 0.0
        for v in x.ravel():
            s += float(v)
        return s

    def sum_list(x):
        s = 0.0
        for v in x:
            s += float(v)
        return s

    def list_to_zeros(x):
        return np.zeros(len(x), dtype=np.float64)

    def _never(*a, **k):  # a Rust sampler reached un-injected would poison a fixture
        raise RuntimeError("un-injected adcraft.rust sampler reached during fixture generation")

    def _validation_only_volume(mean, std):
        # Keyword._validate_volume_sampler calls the sampler once at construction
        # (adcraft/synthetic_kw_classes.py:283-287); the value is discarded.
        return int(round(max(float(mean), 0.0)))

    rust.sum_array_bool = sum_array_bool
    rust.sum_array = sum_array
    rust.sum_list = sum_list
    rust.list_to_zeros = list_to_zeros
    rust.nonneg_int_normal_sampler = _validation_only_volume
    rust.cost_create = _never
    rust.binomial_impressions = _never
    rust.threshold_sigmoid = _never
    rust.repr_outcomes_py = lambda outcomes: ""

    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

        def sample(self):
            return np.zeros(self.shape, dtype=self.dtype)

    class Dict(Space, dict):
        def __init__(self, d):
            dict.__init__(self, d)

        def sample(self):
            return {k: v.sample() for k, v in self.items()}

    class Env:
        _np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    spaces.Space, spaces.Box, spaces.Dict = Space, Box, Dict
    gym.spaces, gym.Env = spaces, Env
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces

    sys.path.insert(0, REF)
    import adcraft  # noqa

    adcraft.rust = rust
    sys.modules["adcraft.rust"] = rust
    return rust


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {path} ({os.path.getsize(path)} B)")


def L(a):
    return np.asarray(a).tolist()


def outcome_lists(outcomes):
    """the per-click lists of the combined BiddingOutcomes (bidding_simulation.py:10-38,124-147), in the reference's order:
    what src/lib.rs:251-275 prints as 'costs' / 'revenues' / 'revenues_per_cost' of info["bidding_outcomes"]"""
    return dict(costs=[[float(x) for x in o["costs"]] for o in outcomes],
                revenues=[[float(x) for x in o["revenues"]] for o in outcomes],
                revenues_per_cost=[[float(x) for x in o["revenues_per_cost"]] for o in outcomes])


def cents(a):
    """2-dp dollars -> integer cents (exact: the reference rounded them to 2 dp)."""
    a = np.asarray(a, dtype=np.float64)
    c = np.rint(a * 100.0)
    assert np.all(np.abs(c / 100.0 - a) < 1e-9), "value is not a whole number of cents"
    return c.astype(np.int64).tolist()


# --------------------------------------------------------------------------- G1
def gen_g1(h):
    rng = np.random.default_rng(101)
    cases = []

    def add(bid, other, n, w, tag):
        imp, pl, co = h.nth_price_auction(bid, np.array(other, dtype=np.float64), n=n, num_winners=w)
        cases.append(dict(tag=tag, bid=float(bid), other_bids=L(other), n=n, num_winners=w,
                          impressions=int(imp), placements=L(pl), costs=L(co)))

    # env path: one competitor, n=2, w=1; ties lose
    add(0.50, [[0.30], [0.50], [0.70], [0.49], [0.51], [0.0]], 2, 1, "env_1competitor_ties")
    for i in range(6):
        nb = 1
        na = int(rng.integers(1, 40))
        other = np.around(np.abs(rng.laplace(0.5, 0.2, (na, nb))), 2)
        add(float(np.around(rng.uniform(0.05, 1.2), 2)), other, 2, 1, f"env_random_{i}")
    # zero auctions
    add(0.5, np.zeros((0, 1)), 2, 1, "zero_auctions")
    # general: many bidders, several n / num_winners, unrounded laplace bids, exact ties
    for i, (nb, n, w) in enumerate([(5, 2, 1), (5, 2, 2), (12, 2, 3), (30, 3, 2), (18, 1, 1), (18, 1, 2),
                                    (2, 2, 2), (3, 3, 3), (1, 3, 1), (4, 2, 2), (7, 4, 1), (25, 2, 1)]):
        na = int(rng.integers(5, 30))
        other = rng.laplace(0.0, 0.1, (na, nb))
        if i % 3 == 0:
            other = np.around(np.abs(other), 2)
        bid = float(np.around(np.abs(rng.laplace(0.15, 0.1)), 2))
        if i % 2 == 0 and na > 2:  # plant exact ties with the bid
            other[1, 0] = bid
            other[2, -1] = bid
        add(bid, other, n, w, f"general_{i}_b{nb}_n{n}_w{w}")
    dump("g1_nth_price_auction.json", dict(
        source="adcraft/synthetic_kw_helpers.py:116-180 (nth_price_auction), executed unmodified",
        cases=cases))


# --------------------------------------------------------------------------- G2
def quant_cfg(eq, mean_volume, cvr, no_vol_prob=None):
    df = eq.generate_simple_experiment_quantiles(mean_volume, cvr)
    # the reference round-trips through CSV (experiment_quantiles.py:66-81); ints become
    # int64 and floats float64 either way, read_csv adds an index column that is never read.
    cfg = {
        "quantiles_folder": "x",  # truthy => load only (gymnasium_kw_utils.py:285-286)
        "load_quant_func": lambda kc: df,
        "mean_volume": mean_volume,
        "conversion_rate": cvr,
    }
    if no_vol_prob is not None:
        cfg["no_vol_prob"] = no_vol_prob
    return cfg, df


def params_to_json(params):
    out = []
    for p in params:
        out.append([[float(p[0][0]), float(p[0][1])]] + [float(x) for x in p[1:]])
    return out


def gen_g2(u, eq):
    cases = []
    table = None
    for seed, K, mv, cvr, nvp in [(10, 30, 100, 0.3, None), (0, 2, 16, 0.5, None), (1729, 64, 128, 0.8, None),
                                  (7, 48, 16, 0.1, 0.5), (3, 16, 64, 0.1, 0.25), (5, 8, 128, 0.8, 1.0)]:
        cfg, df = quant_cfg(eq, mv, cvr, nvp)
        table = {c: L(df[c]) for c in df.columns}
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        kws, params = u.sample_implicit_keywords_from_quantile_dfs(K, rng, cfg)
        cases.append(dict(kind="implicit_quantile", seed=seed, K=K, mean_volume=mv, conversion_rate=cvr,
                          no_vol_prob=nvp, quantile_table=table, params=params_to_json(params),
                          # what the built keyword objects actually hold (scale un-inverted)
                          bctr=[float(k.buyside_ctr) for k in kws],
                          sctr=[float(k.sellside_paid_ctr) for k in kws]))
    dump("g2_keyword_params.json", dict(
        source="adcraft/gymnasium_kw_utils.py:260-349 + pull_quantiles_data/quantiles_to_keywords.py:13-28; "
               "quantile rows from experiment_utils/experiment_quantiles.py:16-47; "
               "param tuple = ((vol_mean, vol_std), cost_loc, 1/cost_scale, bctr, sctr, rev_mean, rev_std)",
        notebook_kat=dict(
            note="printed in adcraft/experiment_utils/example_compute_metrics.ipynb:57,69-76 (seed 10, K 30, "
                 "mean_volume 100, cvr 0.3) and appendix_bidding_outcomes_example/manual_bidding_example.ipynb:84-87 "
                 "(seed 0, K 2, mean_volume 16, cvr 0.5)",
            seed10_kw0=[[100, 13], 0.7386049044669925, 7.3627261809468685, 0.31804044252579394, 0.3,
                        0.9703987841419266, 0.10969622240554196],
            seed0_kw0=[[16, 1], 0.6459721981904619, 9.492169932038324, 0.7526828432972257, 0.5,
                       1.229655446429944, 0.3184237989333203]),
        cases=cases))


def gen_g2_explicit(u, rust):
    # sample_random_keywords builds ExplicitKeyword objects that hold rust.cost_create by
    # reference only (never called at construction), gymnasium_kw_utils.py:90.
    cases = []
    for seed, K in [(1, 10), (0, 64), (1729, 7)]:
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        kws, params = u.sample_random_keywords(K, rng)
        cases.append(dict(kind="explicit_random", seed=seed, K=K, params=params_to_json(params)))
    dump("g2_explicit_params.json", dict(
        source="adcraft/gymnasium_kw_utils.py:113-156 (sample_random_keywords); "
               "param tuple = ((vol_mean, vol_std), imp_intercept, imp_slope, bctr, sctr, rev_mean, rev_std)",
        cases=cases))


# --------------------------------------------------------------------------- G3 (implicit replay traces)
def f32x(x):
    """nearest float32-representable double (so f32 device params hold the same value)."""
    return float(np.float32(x))


def gen_g3(u, b, c):
    traces = []
    spec = [
        # seed, K, mean_volume, cvr, no_vol_prob, budget, bid range
        (11, 6, 16, 0.5, None, 1.0e9, (0.3, 1.0)),
        (12, 12, 128, 0.8, None, 1.0e9, (0.3, 1.0)),
        (13, 8, 16, 0.1, 0.5, 1.0e9, (0.05, 1.5)),
        (14, 10, 128, 0.8, None, 60.0, (0.5, 1.2)),     # binding budget mid-day
        (15, 10, 128, 0.8, None, 5.0, (0.5, 1.2)),      # binding almost immediately
        (16, 5, 64, 0.8, None, 20.0, (0.4, 1.0)),
        (17, 4, 30, 0.5, None, 3.0, (0.6, 1.4)),
        (18, 16, 300, 0.8, None, 1.0e9, (0.2, 0.9)),    # V > 24*... exercises all 24 sub-steps
        (19, 3, 128, 0.8, None, 0.05, (0.5, 1.2)),      # budget smaller than most clicks
        # tie-rich: competitor bids concentrated on 2-3 cent values, budget a multiple of the typical cost, so
        # "remaining == cost" happens and the reference's float residue decides whether the campaign stops
        (20, 6, 64, 0.8, None, 5.0, (0.7, 1.0), (0.50, 0.002)),
        (21, 6, 64, 0.8, None, 2.5, (0.7, 1.0), (0.50, 0.002)),
        (22, 4, 128, 0.8, None, 10.0, (0.6, 0.9), (0.25, 0.001)),
        (23, 8, 40, 0.5, None, 1.0, (0.6, 0.9), (0.10, 0.001)),
        (24, 5, 100, 0.8, None, 7.0, (0.9, 1.2), (0.35, 0.003)),
    ]
    import adcraft.experiment_utils.experiment_quantiles as eq
    for row in spec:
        (seed, K, mv, cvr, nvp, budget, (blo, bhi)) = row[:7]
        override = row[7] if len(row) > 7 else None
        tries = 0
        while True:
            rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed + 1000 * tries)))
            cfg, _ = quant_cfg(eq, mv, cvr, nvp)
            _, params = u.sample_implicit_keywords_from_quantile_dfs(K, rng, cfg)
            # rebuild the keywords with float32-exact parameters so that a device holding f32
            # state sees bit-identical values; volumes come from a recorded sampler
            kws, kp = [], []
            tape = dict(bid=[], click=[], conv=[], rev=[])
            vols = []
            min_margin = [np.inf]
            for p in params:
                vol = (float(p[0][0]), float(p[0][1]))
                loc, scale = f32x(p[1]), f32x(1.0 / p[2])
                if override is not None:
                    loc, scale = f32x(override[0]), f32x(override[1])
                bctr, sctr, mu, sd = f32x(p[3]), f32x(p[4]), f32x(p[5]), f32x(p[6])
                v = int(max(0.0, np.floor(rng.normal(vol[0], max(vol[1], 1e-9)) + 0.5)))
                vols.append(v)
                kw, _p = u.generate_implicit_keyword_from_params(vol, loc, scale, bctr, sctr, mu, sd, rng)
                kw.volume_sampler = (lambda vv: (lambda: vv))(v)
                inner_bid = kw.bid_distribution

                def rec_bid(s, n, _f=inner_bid):
                    out = _f(s, n)
                    tape["bid"].extend(cents(out.reshape(-1)))
                    return out
                kw.bid_distribution = rec_bid
                inner_rev = kw.reward_distribution_sampler

                def rec_rev(n, _f=inner_rev):
                    out = _f(n)
                    tape["rev"].extend(cents(out))
                    return out
                kw.reward_distribution_sampler = rec_rev
                kws.append(kw)
                kp.append(dict(vol_mean=vol[0], vol_std=vol[1], loc=loc, scale=scale, bctr=bctr, sctr=sctr,
                               rev_mean=mu, rev_std=sd))
            bids = [float(x) for x in np.around(rng.uniform(blo, bhi, K), 2)]

            # coinflips is looked up as a module global of synthetic_kw_classes
            # (adcraft/synthetic_kw_classes.py:219,233): wrap it to record the booleans in call order.
            orig_coin = c.coinflips
            calls = []

            def rec_coin(p, n, rng_):
                out = orig_coin(p, n, rng_)
                calls.append(out.copy())
                return out
            c.coinflips = rec_coin
            # record the tie margin of every budget check the reference makes: the engine
            # works in exact integer cents, the reference in binary floating point, so an
            # exact tie (remaining == cost) may resolve either way there.
            try:
                outcomes = b.simulate_epoch_of_bidding_on_campaign(kws, bids, budget)
            finally:
                c.coinflips = orig_coin
            # coinflip calls alternate click, conv per visited cell
            for i, arr in enumerate(calls):
                tape["click" if i % 2 == 0 else "conv"].extend([int(x) for x in arr])
            # exact ties (remaining == cost, or the campaign's remaining hitting exactly zero) are KEPT: the
            # restatement reproduces the reference's float arithmetic there; has_tie is recorded for information
            has_tie = not tie_free(vols, bids, budget, tape, K)
            break
        traces.append(dict(
            seed=seed, tries=tries, has_tie=has_tie, K=K, budget=budget, bids=bids, volumes=vols, keyword_params=kp,
            tape=tape,
            out=dict(impressions=[int(o["impressions"]) for o in outcomes],
                     buyside_clicks=[int(o["buyside_clicks"]) for o in outcomes],
                     sellside_conversions=[int(o["sellside_conversions"]) for o in outcomes],
                     cost=[float(np.sum(np.asarray(o["costs"], dtype=np.float64))) if len(o["costs"]) else 0.0
                           for o in outcomes],
                     revenue=[float(np.sum(np.asarray(o["revenues"], dtype=np.float64))) if len(o["revenues"]) else 0.0
                              for o in outcomes],
                     profit=[float(o["profit"]) for o in outcomes],
                     impression_share=[float(o["impression_share"]) for o in outcomes],
                     **outcome_lists(outcomes))))
    dump("g3_implicit_replay.json", dict(
        source="adcraft/bidding_simulation.py:170-234 (simulate_epoch_of_bidding_on_campaign) over "
               "ImplicitKeyword objects from gymnasium_kw_utils.py:169-195, executed unmodified; "
               "tapes are the variates the reference drew, in call order (t-major, kw-minor): "
               "bid = competitor bids in cents (n per visited cell), click = booleans (one per won auction), "
               "conv = booleans (one per paid click), rev = revenues in cents (one per conversion); "
               "out.costs / revenues / revenues_per_cost = the combined outcomes' per-click lists per keyword in the reference's order, "
               "out.impression_share / profit as combine_outcomes leaves them (bidding_simulation.py:124-147)",
        traces=traces))


def tie_free(vols, bids, budget, tape, K):
    """Exact-cents replay of the budget walk; False if any check has remaining == cost > 0
    or the campaign-level remaining hits exactly zero (float noise decides those in the reference)."""
    rem = int(round(budget * 100))
    bc = [int(round(x * 100)) for x in bids]
    ib = ic = 0
    step = [v // 24 for v in vols]
    for t in range(24):
        for k in range(K):
            n = vols[k] - 23 * step[k] if t == 0 else step[k]
            comp = tape["bid"][ib:ib + n]
            ib += n
            wins = [x for x in comp if bc[k] > x]
            clicks = tape["click"][ic:ic + len(wins)]
            ic += len(wins)
            spent = 0
            r = rem
            for cl, co in zip(clicks, wins):
                if cl:
                    if r == co and co > 0:
                        return False
                    if r >= co:
                        r -= co
                        spent += co
                    else:
                        break
            rem -= spent
            if rem == 0 and budget < 1e8:
                return False
            if rem <= 0:
                return True
    return True


# --------------------------------------------------------------------------- G3b (explicit replay traces)
def gen_g3_explicit(u, b, c, rust):
    traces = []
    for (seed, K, budget, bidhi) in [(21, 5, 1000.0, 1.5), (22, 8, 1000.0, 0.6), (23, 6, 12.0, 1.5), (24, 4, 3.0, 2.0),
                                     (25, 10, 1.0e9, 1.0)]:
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        _, params = u.sample_random_keywords(K, rng)
        tape = dict(impressions=[], p=[], cost=[], click=[], conv=[], rev=[])
        kws, kp, vols = [], [], []
        aux = np.random.default_rng(seed + 7)

        # recorded stand-ins for the three Rust calls on the EXPLICIT path
        def thr_sig(x, prm):  # formula of src/lib.rs:93-105,290-300 (deterministic; parity for it is
            # pinned separately by closed form, see tests/test_oracle_scalar.py)
            halver = 2.0 + 1e-10
            th = min(max(halver * prm["impression_thresh"], 0.0), 1.0) / halver
            r = 1.0 / (1.0 + np.exp(-prm["impression_slope"] * (x - prm["impression_bid_intercept"])))
            return float(min(max((1.0 + 2.0 * th) * r - th, 0.0), 1.0))

        def binom(n, p):
            v = int(aux.binomial(n, p))
            tape["impressions"].append(v)
            tape["p"].append(float(p))
            return v

        def cost_create(x, n):
            sq = np.sqrt(x)
            out = np.clip(sq / 4.0 + 2.2 + aux.normal(0.0, 1e-10 + sq / 6.0, n), 0.0, 4.4)
            tape["cost"].extend(float(v) for v in out)
            return out
        rust.threshold_sigmoid = thr_sig
        rust.binomial_impressions = binom
        rust.cost_create = cost_create
        for p in params:
            vol = (float(p[0][0]), float(p[0][1]))
            ii, sl = f32x(p[1]), f32x(p[2])
            bctr, sctr, mu, sd = f32x(p[3]), f32x(p[4]), f32x(p[5]), f32x(p[6])
            v = int(max(0.0, np.floor(aux.normal(vol[0], max(vol[1], 1e-9)) + 0.5)))
            vols.append(v)
            kw, _ = u.generate_keyword_from_params(vol, ii, sl, bctr, sctr, mu, sd, rng)
            kw.volume_sampler = (lambda vv: (lambda: vv))(v)
            kw.cost_per_buyside_click = cost_create
            inner_rev = kw.reward_distribution_sampler

            def rec_rev(n, _f=inner_rev):
                out = _f(n)
                tape["rev"].extend(cents(out))
                return out
            kw.reward_distribution_sampler = rec_rev
            kws.append(kw)
            kp.append(dict(vol_mean=vol[0], vol_std=vol[1], imp_intercept=ii, imp_slope=sl, imp_thresh=0.05,
                           bctr=bctr, sctr=sctr, rev_mean=mu, rev_std=sd))
        bids = [float(x) for x in np.around(aux.uniform(0.05, bidhi, K), 2)]
        orig_coin = c.coinflips
        calls = []

        def rec_coin(p, n, rng_):
            out = orig_coin(p, n, rng_)
            calls.append(out.copy())
            return out
        c.coinflips = rec_coin
        try:
            outcomes = b.simulate_epoch_of_bidding_on_campaign(kws, bids, budget)
        finally:
            c.coinflips = orig_coin
            rust.threshold_sigmoid = rust.binomial_impressions = rust.cost_create = None
        for i, arr in enumerate(calls):
            tape["click" if i % 2 == 0 else "conv"].extend([int(x) for x in arr])
        traces.append(dict(
            seed=seed, K=K, budget=budget, bids=bids, volumes=vols, keyword_params=kp, tape=tape,
            out=dict(impressions=[int(o["impressions"]) for o in outcomes],
                     buyside_clicks=[int(o["buyside_clicks"]) for o in outcomes],
                     sellside_conversions=[int(o["sellside_conversions"]) for o in outcomes],
                     cost=[float(rust_sum_list(o["costs"])) for o in outcomes],
                     revenue=[float(rust_sum_list(o["revenues"])) for o in outcomes],
                     profit=[float(o["profit"]) for o in outcomes],
                     impression_share=[float(o["impression_share"]) for o in outcomes],
                     **outcome_lists(outcomes))))
    dump("g3_explicit_replay.json", dict(
        source="adcraft/bidding_simulation.py:170-234 over ExplicitKeyword objects "
               "(adcraft/synthetic_kw_classes.py:457-575) built by gymnasium_kw_utils.py:67-96, executed unmodified; "
               "the three Rust samplers on this path are replaced by recorded injected samplers: "
               "tape.impressions/p = Binomial result and its p per visited cell, tape.cost = per-impression costs "
               "(the phantom [0] of synthetic_kw_classes.py:514-515 is produced by the reference itself and is NOT "
               "on the tape), click/conv booleans and revenues in cents as in g3_implicit_replay; out.costs / revenues / "
               "revenues_per_cost / impression_share: the combined outcomes' per-click lists, as in g3_implicit_replay",
        traces=traces))


def rust_sum_list(x):
    s = 0.0
    for v in x:
        s += float(v)
    return s


# --------------------------------------------------------------------------- G4 (drift)
def gen_g4(env_mod, eq):
    seqs = []
    for seed, K, mv, cvr, up in [(31, 6, 128, 0.8, [["vol", 0.03], ["ctr", 0.03], ["cvr", 0.03]]),
                                 (32, 4, 16, 0.1, [["vol", 0.1], ["ctr", 0.5], ["cvr", 0.9]])]:
        cfg, _ = quant_cfg(eq, mv, cvr)
        env = env_mod.BiddingSimulation(keyword_config=cfg, num_keywords=K, updater_params=up,
                                        updater_mask=[True] * K)
        env.reset(seed=seed)
        p0 = params_to_json(env.keyword_params)
        # record the uniforms by drawing them from a clone of the generator state
        steps = []
        for t in range(5):
            st = env.np_random.bit_generator.state
            clone = np.random.Generator(np.random.PCG64())
            clone.bit_generator.state = st
            draws = [L(clone.uniform(-v[1], v[1], size=(K,))) for v in up]
            env.update_keywords()
            steps.append(dict(uniforms=draws, params=params_to_json(env.keyword_params),
                              kw_bctr=[float(k.buyside_ctr) for k in env.keywords],
                              kw_sctr=[float(k.sellside_paid_ctr) for k in env.keywords]))
        seqs.append(dict(seed=seed, K=K, mean_volume=mv, conversion_rate=cvr, updater_params=up,
                         params0=p0, steps=steps))
    dump("g4_update_keywords.json", dict(
        source="adcraft/gymnasium_kw_env.py:114-158 (update_keywords) on an env reset with a seed, executed unmodified",
        sequences=seqs))


# --------------------------------------------------------------------------- G5 (metrics)
def gen_g5(m, u, eq):
    rng = np.random.default_rng(55)
    cases = []
    for T, K in [(60, 10), (5, 3), (1, 1), (7, 16)]:
        prof = rng.normal(2.0, 5.0, (T, K))
        ideal = rng.normal(3.0, 4.0, (T, K))
        if K > 2:
            ideal[:, 1] = -1.0
        cases.append(dict(kw_profits=L(prof), ideal_profits=L(ideal),
                          AKNCP=float(m.compute_AKNCP(prof, ideal)), NCP=float(m.compute_NCP(prof, ideal))))
    cases.append(dict(kw_profits=L(np.ones((3, 2))), ideal_profits=L(-np.ones((3, 2))),
                      AKNCP=float(m.compute_AKNCP(np.ones((3, 2)), -np.ones((3, 2)))),
                      NCP=float(m.compute_NCP(np.ones((3, 2)), -np.ones((3, 2))))))
    maxp = []
    for i in range(6):
        nb = 299
        kwp = [[int(rng.integers(1, 200)), 3.0], 0.5, 5.0, float(rng.uniform(0.05, 0.9)), float(rng.uniform(0.05, 0.9)),
               float(rng.uniform(0.2, 1.5)), 0.1]
        cpc = np.sort(rng.uniform(0.0, 1.0, nb))
        ir = np.sort(rng.uniform(0.0, 1.0, nb))
        if i == 5:
            cpc = cpc + 5.0  # never profitable
        r = m.get_max_expected_bid_profits(kwp, cpc, ir)
        maxp.append(dict(kw_params=kwp, cpc=L(cpc), ir=L(ir), max_profit=float(r[0]), frac_positive=float(r[1]),
                         argmax=int(r[2])))
    # get_implicit_kw_bid_cpc_impressions: feed a keyword whose sample_bids is a recorded array
    curves = []
    bid_array = np.arange(0.01, 3.00, 0.01)
    for i in range(3):
        n = [2048, 64, 2048][i]
        samples = np.around(np.abs(rng.laplace(0.55, 0.08 + 0.05 * i, (1, n))), 2)

        class _KW:
            def sample_bids(self, k, _s=samples):
                assert k == _s.shape[1]
                return _s
        ir, cpc = m.get_implicit_kw_bid_cpc_impressions(_KW(), bid_array, n_samples=n)
        curves.append(dict(samples_cents=cents(samples.reshape(-1)), n_samples=n, impression_rates=L(ir), cpc=L(cpc)))
    dump("g5_metrics.json", dict(
        source="adcraft/experiment_utils/experiment_metrics.py:20-83, executed unmodified; "
               "bid_array = np.arange(0.01, 3.00, 0.01) as in the notebooks",
        bid_array=L(bid_array), akncp_ncp=cases, max_expected=maxp, bid_curves=curves))


# --------------------------------------------------------------------------- G6 / G7
def gen_g6(u):
    rng = np.random.default_rng(66)
    K = 5
    obs = dict(impressions=rng.integers(0, 50, K), buyside_clicks=rng.integers(0, 20, K),
               cost=rng.uniform(0, 9, K), sellside_conversions=rng.integers(0, 9, K),
               revenue=rng.uniform(0, 9, K), cumulative_profit=np.array([12.5]), days_passed=np.array([3]))
    flat = u.flatten_dict_array(obs)
    dump("g6_flatten.json", dict(
        source="adcraft/gymnasium_kw_utils.py:383-390 (flatten_dict_array)",
        obs={k: L(v) for k, v in obs.items()}, flat=L(flat), key_order=sorted(obs.keys())))


def gen_g7(h):
    # tables held by the reference's own tests (data, transcribed), plus reference outputs
    # of the pure-numpy samplers under a fixed PCG64 seed.
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(77)))
    lap = h.bid_abs_laplace(0.55, 0.08, rng)(1, 32)
    rev = h.rev_normal(1.0, 0.15, rng)(32)
    coin = h.coinflips(0.3, 32, rng)
    rng2 = np.random.Generator(np.random.PCG64(np.random.SeedSequence(78)))
    nor = h.bid_abs_normal(0.4, 0.2, rng2, 0.05)(2, 8)
    dump("g7_kat_tables.json", dict(
        sigmoid=dict(source="adcraft/tests/test_synthetic_kw_helpers.py:72-82 (rounded to 4 dp)",
                     rows=[[0, 0, 0, 0.5], [1, 0, 0, 0.5], [1, 1, 0, 0.7311], [1, 1, 1, 0.5], [-1, 1, 0, 0.2689],
                           [-10, 1, 0, 0.0], [1, -1, 0, 0.2689], [1, -10, 0, 0.0]]),
        probify=dict(source="adcraft/tests/test_synthetic_kw_helpers.py:14-21",
                     rows=[[0, 0], [1, 1], [2, 1], [-1, 0], [0.5, 0.5]],
                     array=[[0, 1, 2, -1, 0.5], [0, 1, 1, 0, 0.5]]),
        nonnegify=dict(source="adcraft/tests/test_synthetic_kw_helpers.py:34-41",
                       rows=[[0, 0], [1, 1], [2, 2], [-1, 0], [0.5, 0.5]],
                       array=[[0, 1, 2, -1, 0.5], [0, 1, 2, 0, 0.5]]),
        beta_param=dict(source="adcraft/tests/test_synthetic_kw_helpers.py:54-67", rows=[[2, -0.5], [-1, -2], [0.5, 1]]),
        sum_array_bool=dict(source="adcraft/tests/rust/test_numpy_funcs.py:13-17",
                            rows=[[[True, True, True, True], 4], [[False, False, False, False], 0],
                                  [[True, False, True, False], 2]]),
        sum_array=dict(source="adcraft/tests/rust/test_numpy_funcs.py:71-75",
                       rows=[[[1.0, 1.0, 1.0, 1.0], 4], [[0.0, 0.0, 0.0, 0.0], 0], [[1.0, 0.0, 1.0, 0.0], 2]]),
        sum_list=dict(source="adcraft/tests/rust/test_numpy_funcs.py:101-118",
                      rows=[[[1, 1, 1, 1], 4], [[1.0, 1.0, 1.0, 1.0], 4], [[0, 0, 0, 0], 0], [[1, 0, 1, 0], 2],
                            [[True, True, False, False], 2],
                            [[44.7, 88.465, 38.462, 300.0], float(np.array([44.7, 88.465, 38.462, 300.0]).sum())]]),
        probify_float=dict(source="adcraft/tests/rust/test_numpy_funcs.py:139-151",
                           rows=[[0.0, 0.0, 1.0], [-10.0, 0.0, 1.0], [10.0, 0.0, 1.0], [0.5, 0.0, 1.0], [0.4999, 0.0, 1.0]]),
        seeded_samplers=dict(
            source="adcraft/synthetic_kw_helpers.py:66-77,92-113 executed on Generator(PCG64(SeedSequence(77|78)))",
            bid_abs_laplace=dict(seed=77, loc=0.55, scale=0.08, shape=[1, 32], out=L(lap)),
            rev_normal=dict(mean=1.0, std=0.15, n=32, out=L(rev)),
            coinflips=dict(p=0.3, n=32, out=[int(x) for x in coin]),
            bid_abs_normal=dict(seed=78, loc=0.4, scale=0.2, lowest_bid=0.05, shape=[2, 8], out=L(nor)))))


# --------------------------------------------------------------------------- G8 (whole step() episodes)
def gen_g8(env_mod, eq, c, rust):
    eps = []
    todo = [(41, 5, 32, 0.8, 1000.0, 10000.0, 4, False, 0),
            (42, 6, 64, 0.8, 8.0, 30.0, 6, True, 0),
            (43, 4, 128, 0.8, 1000.0, 10000.0, 3, True, 0),
            (44, 5, 64, 0.1, 1000.0, 20.0, 10, False, 0)]  # ends by truncation (loss threshold)
    while todo:
        (seed, K, mv, cvr, budget, loss_thr, max_days, drift, tries) = todo.pop(0)
        assert tries < 50
        cfg, _ = quant_cfg(eq, mv, cvr)
        env = env_mod.BiddingSimulation(keyword_config=cfg, num_keywords=K, budget=budget, loss_threshold=loss_thr,
                                        max_days=max_days, updater_mask=[True] * K if drift else None)
        env.reset(seed=seed)
        # f32-exact parameters + recorded samplers (same method as G3)
        aux = np.random.default_rng(seed + 3 + 1000 * tries)
        tape = dict(bid=[], click=[], conv=[], rev=[])
        vol_log = []

        def rewire():
            for k, (kw, p) in enumerate(zip(env.keywords, env.keyword_params)):
                def vs(_p=p):
                    v = int(max(0.0, np.floor(aux.normal(_p[0][0], max(_p[0][1], 1e-9)) + 0.5)))
                    vol_log.append(v)
                    return v
                kw.volume_sampler = vs
        for kw, p in zip(env.keywords, env.keyword_params):
            loc, scale = f32x(p[1]), f32x(1.0 / p[2])
            p[1], p[2] = loc, 1.0 / scale
            p[3] = kw.buyside_ctr = f32x(p[3])
            p[4] = kw.sellside_paid_ctr = f32x(p[4])
            p[5], p[6] = f32x(p[5]), f32x(p[6])
            import adcraft.synthetic_kw_helpers as h
            base_bid = h.bid_abs_laplace(loc, scale, env.np_random)
            base_rev = h.rev_normal(p[5], p[6], env.np_random)

            def rec_bid(s, n, _f=base_bid):
                out = _f(s, n)
                tape["bid"].extend(cents(out.reshape(-1)))
                return out

            def rec_rev(n, _f=base_rev):
                out = _f(n)
                tape["rev"].extend(cents(out))
                return out
            kw.bid_distribution = rec_bid
            kw.reward_distribution_sampler = rec_rev
        params0 = params_to_json(env.keyword_params)
        # update_keywords rebuilds volume_sampler through the rust wrapper each step
        # (gymnasium_kw_env.py:150-152): route that wrapper to the recorded sampler too.
        rust.nonneg_int_normal_sampler = lambda mean, std: (
            vol_log.append(int(max(0.0, np.floor(aux.normal(mean, max(std, 1e-9)) + 0.5)))) or vol_log[-1])
        rewire()
        orig_coin = c.coinflips
        calls = []

        def rec_coin(p, n, rng_):
            out = orig_coin(p, n, rng_)
            calls.append(out.copy())
            return out
        c.coinflips = rec_coin
        # step() hands its combined outcomes to rust.repr_outcomes_py (gymnasium_kw_env.py:249): keep what it was given
        seen = []
        rust.repr_outcomes_py = lambda outcomes: (seen.append(dict(
            impression_share=[float(o["impression_share"]) for o in outcomes], profit=[float(o["profit"]) for o in outcomes],
            **outcome_lists(outcomes))), "")[1]
        steps = []
        try:
            for t in range(max_days):
                bids = np.around(aux.uniform(0.3, 1.1, K), 2)
                ntape0 = {k: len(v) for k, v in tape.items()}
                ncalls0, nvol0 = len(calls), len(vol_log)
                drift_u = None
                if drift:
                    st = env.np_random.bit_generator.state  # state BEFORE the step; drift draws come last
                obs, rew, term, trunc, info = env.step({"keyword_bids": bids, "budget": budget})
                for i, arr in enumerate(calls[ncalls0:]):
                    tape["click" if i % 2 == 0 else "conv"].extend([int(x) for x in arr])
                steps.append(dict(
                    bids=L(bids), budget=budget, volumes=vol_log[nvol0:nvol0 + K],
                    tape_slices={k: [ntape0[k], len(tape[k])] for k in tape},
                    obs={k: L(v) for k, v in obs.items()}, reward=float(rew), terminated=bool(term),
                    truncated=bool(trunc), params_after=params_to_json(env.keyword_params), outcomes=seen[-1]))
                if term or trunc:
                    break
        finally:
            c.coinflips = orig_coin
            rust.repr_outcomes_py = lambda outcomes: ""
        ok = True
        for st_ in steps:
            sl = st_["tape_slices"]
            ok = ok and tie_free(st_["volumes"], st_["bids"], budget,
                                 {k: tape[k][sl[k][0]:sl[k][1]] for k in tape}, K)
        eps.append(dict(seed=seed, tries=tries, has_tie=not ok, K=K, budget=budget, loss_threshold=loss_thr, max_days=max_days,
                        drift=drift, params0=params0, tape=tape, steps=steps))
    dump("g8_env_episodes.json", dict(
        source="adcraft/gymnasium_kw_env.py:160-269 (BiddingSimulation.step) executed unmodified over recorded "
               "samplers; obs/reward/terminated/truncated are the reference's; params_after shows the drift of "
               "gymnasium_kw_env.py:114-158 (its uniforms are NOT on the tape: the engine's drift stream is its own; "
               "drift arithmetic is pinned by g4); steps[].outcomes = the combined BiddingOutcomes step() passed to "
               "rust.repr_outcomes_py (:249): per-click lists per keyword, impression_share, profit",
        episodes=eps))


def main():
    rust = install_standins()
    from adcraft import synthetic_kw_helpers as h, synthetic_kw_classes as c, bidding_simulation as b
    from adcraft import gymnasium_kw_utils as u, gymnasium_kw_env as env_mod
    from adcraft.experiment_utils import experiment_metrics as m, experiment_quantiles as eq
    gen_g1(h)
    gen_g2(u, eq)
    gen_g2_explicit(u, rust)
    gen_g3(u, b, c)
    gen_g3_explicit(u, b, c, rust)
    gen_g4(env_mod, eq)
    gen_g5(m, u, eq)
    gen_g6(u)
    gen_g7(h)
    gen_g8(env_mod, eq, c, rust)


if __name__ == "__main__":
    main()
