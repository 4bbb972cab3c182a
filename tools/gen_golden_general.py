#!/usr/bin/env python3
"""G12: replay traces of the reference's DEFAULT ImplicitKeyword (the non-env model of SURVEY 8a row a19) in its own
campaign loop - bidders B ~ Binomial(max_bidders, participation_rate) drawn once per (sub-timestep, keyword) call, raw
Laplace(bid_loc, bid_scale) bids of every bidder in every auction, the literal top-(w+n) second-price clearing of
nth_price_auction (adcraft/synthetic_kw_classes.py:610-686, adcraft/synthetic_kw_helpers.py:116-180), float64 money.

Runs ONLY in the build container (imports the reference from /root/reference with the in-memory stand-ins of
tools/gen_golden.py); writes tests/golden/g12_implicit_general_replay.json: the variates the reference drew, in its order,
and the outcomes it computed.

    python tools/gen_golden_general.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (stand-ins, dump, cents)


def main():
    G.install_standins()
    sys.path.insert(0, G.REF)
    import adcraft.bidding_simulation as b
    import adcraft.synthetic_kw_classes as c
    import adcraft.synthetic_kw_helpers as h

    spec = [
        # seed, K, volumes ~ U[lo, hi], budget, bid range, max_bidders, participation, bid_loc, bid_scale
        (41, 5, (0, 60), 1.0e9, (0.05, 0.45), 30, 0.6, 0.0, 0.1),          # the reference's defaults
        (42, 5, (20, 120), 1.0e9, (0.10, 0.60), 30, 0.6, 0.0, 0.1),
        (43, 4, (30, 90), 2.0, (0.20, 0.60), 30, 0.6, 0.0, 0.1),           # binding budget
        (44, 4, (10, 80), 0.5, (0.25, 0.50), 30, 0.6, 0.0, 0.1),           # binding almost at once
        (45, 6, (0, 90), 1.0e9, (0.02, 0.30), 4, 0.3, 0.0, 0.1),           # often fewer than w+n bidders: zero padding
        (46, 5, (10, 100), 20.0, (0.30, 0.90), 12, 0.5, 0.35, 0.08),       # another competitor law
        (47, 2, (600, 700), 1.0e9, (0.15, 0.40), 8, 0.5, 0.0, 0.1),        # more than 24 auctions per sub-timestep
    ]
    traces = []
    for seed, K, (vlo, vhi), budget, (blo, bhi), max_bidders, rate, bid_loc, bid_scale in spec:
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        tape = dict(bidders=[], bids=[], click=[], conv=[], rev=[])
        kws, kp, vols = [], [], []
        for _ in range(K):
            bctr, sctr = G.f32x(rng.uniform(0.2, 0.9)), G.f32x(rng.uniform(0.1, 0.9))
            mu, sd = G.f32x(rng.uniform(0.3, 1.5)), G.f32x(rng.uniform(0.02, 0.3))
            loc, scale = G.f32x(bid_loc), G.f32x(bid_scale)
            v = int(rng.integers(vlo, vhi + 1))
            kw = c.ImplicitKeyword({"rng": rng, "volume": v, "buyside_ctr": bctr, "sellside_paid_ctr": sctr,
                                    "reward_distribution_sampler": h.rev_normal(mu, sd, rng),
                                    "max_bidders": max_bidders, "participation_rate": rate, "bid_loc": loc, "bid_scale": scale})
            inner_n = kw.bidder_distribution            # the reference's own default samplers, recorded as they are called

            def rec_n(_f=inner_n):
                out = _f()
                tape["bidders"].append(int(out))
                return out
            kw.bidder_distribution = rec_n
            inner_b = kw.bid_distribution

            def rec_b(s, n, _f=inner_b):
                out = _f(s, n)                      # shape (bidders, auctions)
                tape["bids"].extend(float(x) for x in out.reshape(-1))
                return out
            kw.bid_distribution = rec_b
            inner_r = kw.reward_distribution_sampler

            def rec_r(n, _f=inner_r):
                out = _f(n)
                tape["rev"].extend(G.cents(out))
                return out
            kw.reward_distribution_sampler = rec_r
            kws.append(kw)
            vols.append(v)
            kp.append(dict(bctr=bctr, sctr=sctr, rev_mean=mu, rev_std=sd, bid_loc=loc, bid_scale=scale))
        bids = [float(x) for x in np.around(rng.uniform(blo, bhi, K), 2)]
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
        for i, arr in enumerate(calls):          # coinflip calls alternate click, conv per visited cell
            tape["click" if i % 2 == 0 else "conv"].extend([int(x) for x in arr])
        traces.append(dict(
            seed=seed, K=K, budget=budget, bids=bids, volumes=vols, max_bidders=max_bidders, participation_rate=rate,
            keyword_params=kp, tape=tape,
            out=dict(impressions=[int(o["impressions"]) for o in outcomes],
                     buyside_clicks=[int(o["buyside_clicks"]) for o in outcomes],
                     sellside_conversions=[int(o["sellside_conversions"]) for o in outcomes],
                     cost=[G.rust_sum_list(o["costs"]) for o in outcomes],
                     revenue=[float(np.sum(np.asarray(o["revenues"], dtype=np.float64))) if len(o["revenues"]) else 0.0 for o in outcomes],
                     profit=[float(o["profit"]) for o in outcomes],
                     impression_share=[float(o["impression_share"]) for o in outcomes], **G.outcome_lists(outcomes))))
        print(f"seed {seed}: {sum(vols)} auctions, {len(tape['bidders'])} cells visited, {len(tape['bids'])} bids, "
              f"impressions {sum(traces[-1]['out']['impressions'])}, clicks {sum(traces[-1]['out']['buyside_clicks'])}")
    G.dump("g12_implicit_general_replay.json", dict(
        source="adcraft/bidding_simulation.py:170-234 over DEFAULT ImplicitKeyword objects (adcraft/synthetic_kw_classes.py:578-688: "
               "bidders ~ Binomial(max_bidders, participation_rate) once per (t, keyword) call, raw Laplace(bid_loc, bid_scale) "
               "bids, nth_price_auction with n=2, num_winners=1), executed unmodified; tapes in call order (t-major, keyword-"
               "minor): bidders = one count per visited cell; bids = float64 bids, bidders x auctions per cell, bidder-major as "
               "bid_distribution(s, n) returns them; click = one boolean per won auction; conv = one per paid click; rev = "
               "revenue in cents per conversion; out.costs / revenues / revenues_per_cost / impression_share: the combined outcomes' "
               "per-click lists and combine_outcomes' impression_share (bidding_simulation.py:10-38,124-147), as in g3_implicit_replay",
        traces=traces))


if __name__ == "__main__":
    main()
