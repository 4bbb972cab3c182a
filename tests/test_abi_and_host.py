"""-m "not gpu": the C-ABI library loads and exports every symbol include/adcraft_engine.h declares; host-side
logic (keyword generation, spaces, adcraft.rust shim, metrics, flattening) against the golden vectors.
No engine compute happens here (there is no GPU); creating an engine must fail loudly, not fall back."""
import os
import re
import subprocess

import numpy as np
import pytest

from adcraft_amd import _ffi, experiment_metrics as em, gymnasium_kw_utils as utils, rust, spaces
from adcraft_amd.gymnasium_kw_env import BiddingSimulation, bidding_sim_creator
from adcraft_amd.wrappers import FlatArrayWrapper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rng(seed):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "adcraft_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 40
    lib = _ffi.lib()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.library_path()], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in names:
        assert n in exported, f"{n} declared in include/adcraft_engine.h but not exported"
        getattr(lib, n)
    assert lib.adc_abi_version() == 5 and lib.adc_stream_revision() == 5
    # nothing but the ABI is exported (kernels and helpers stay hidden)
    assert all(e.startswith("adc_") for e in exported if not e.startswith("_"))


def test_no_cpu_fallback_engine_creation_fails_loudly_without_gpu():
    if _ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    from adcraft_amd.engine import StepEngine
    with pytest.raises(_ffi.EngineError, match="no CPU path"):
        StepEngine(1, 4)
    env = BiddingSimulation(num_keywords=4)
    with pytest.raises(_ffi.EngineError):
        env.reset(seed=1)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "adcraft_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt and "adcraft_oracle" not in txt, f


# ------------------------------------------------------------------ keyword generation == reference (G2)
def test_implicit_keyword_params_bit_exact(golden):
    for c in golden("g2_keyword_params.json")["cases"]:
        cfg = utils.experiment_keyword_config(c["mean_volume"], c["conversion_rate"])
        if c["no_vol_prob"] is not None:
            cfg["no_vol_prob"] = c["no_vol_prob"]
        raw = utils.sample_implicit_keyword_params(c["K"], _rng(c["seed"]), cfg)
        printed = utils.printed_params(raw, implicit=True)
        got = [[[float(p[0][0]), float(p[0][1])]] + [float(x) for x in p[1:]] for p in printed]
        assert got == c["params"]


def test_explicit_keyword_params_bit_exact(golden):
    for c in golden("g2_explicit_params.json")["cases"]:
        raw = utils.sample_random_keyword_params(c["K"], _rng(c["seed"]))
        got = [[[float(p[0][0]), float(p[0][1])]] + [float(x) for x in p[1:]] for p in raw]
        assert got == c["params"]


def test_construction_draws_line_up_with_the_reference(golden):
    """after keyword generation the env generator must be where the reference's is: the first drift uniforms match"""
    s = golden("g4_update_keywords.json")["sequences"][0]
    rng = _rng(s["seed"])
    utils.sample_implicit_keyword_params(s["K"], rng, utils.experiment_keyword_config(s["mean_volume"], s["conversion_rate"]))
    utils.consume_construction_draws(s["K"], rng)
    draws = [rng.uniform(-v[1], v[1], size=(s["K"],)).tolist() for v in s["updater_params"]]
    assert draws == s["steps"][0]["uniforms"]


def test_quantile_table_equals_reference(golden):
    c = golden("g2_keyword_params.json")["cases"][0]
    t = utils.generate_simple_experiment_quantiles(c["mean_volume"], c["conversion_rate"])
    assert {k: [float(x) for x in v] for k, v in t.items()} == {k: [float(x) for x in v] for k, v in c["quantile_table"].items()}


def test_planes_layout():
    raw = utils.sample_implicit_keyword_params(5, _rng(3), utils.experiment_keyword_config(16, 0.1, no_vol_prob=0.5))
    pl = utils.implicit_params_to_planes(raw)
    assert pl.shape == (8, 5) and pl.dtype == np.float32
    for k, p in enumerate(raw):
        assert pl[0, k] == np.float32(p[0][0]) and pl[3, k] == np.float32(p[2]) and pl[7, k] == np.float32(p[6])


# ------------------------------------------------------------------ env API shape (adcraft/tests/test_env.py)
def test_env_api_surface():
    env = BiddingSimulation()
    assert hasattr(env, "observation_space") and hasattr(env, "action_space")
    assert env.num_keywords == 10 and env.budget == 1000.0 and env.max_days == 60 and env.loss_threshold == 10000.0
    assert env.metadata == {"render_modes": ["ansi"]}
    assert set(env.action_space.keys()) == {"keyword_bids", "budget"}
    assert set(env.observation_space.keys()) == {"impressions", "buyside_clicks", "cost", "sellside_conversions",
                                                 "revenue", "cumulative_profit", "days_passed"}
    assert env.action_space["keyword_bids"].shape == (10,) and env.action_space["keyword_bids"].dtype == np.float32
    env2 = bidding_sim_creator(dict(num_keywords=3, config={}))      # unknown kwargs are swallowed (multi_agent/env.py:31)
    assert env2.num_keywords == 3
    with pytest.raises(AssertionError):
        BiddingSimulation(render_mode="human")                       # gymnasium_kw_env.py:91-93
    with pytest.raises(AssertionError):
        env.set_updater_mask([True] * 3)                             # :107-110
    with pytest.raises(AssertionError):
        env.step({"keyword_bids": np.ones(10, np.float32)})          # :194-196 reset required
    w = FlatArrayWrapper(env)
    assert w.observation_space.shape == (52,) and w.action_space.shape == (11,)
    a = w.action(np.arange(11, dtype=np.float32))
    assert a["budget"].tolist() == [0.0] and a["keyword_bids"].tolist() == list(range(1, 11))


def test_spaces_contains_and_sample():
    sp = spaces.get_observation_space(4, 100.0)
    zero = dict(impressions=np.zeros(4, int), buyside_clicks=np.zeros(4, int), cost=np.zeros(4, np.float32),
                sellside_conversions=np.zeros(4, int), revenue=np.zeros(4, np.float32),
                cumulative_profit=np.zeros(1, np.float32), days_passed=np.zeros(1, np.float32))
    assert sp.contains(zero)
    bad = dict(zero, cost=np.full(4, 101.0, np.float32))
    assert not sp.contains(bad)
    a = spaces.get_action_space(4).sample()
    assert a["keyword_bids"].shape == (4,) and (a["keyword_bids"] >= 0.01).all()


# ------------------------------------------------------------------ adcraft.rust shim (adcraft/tests/rust/*)
def test_rust_shim_reducers_and_type_errors(golden):
    g = golden("g7_kat_tables.json")
    for x, exp in g["sum_array_bool"]["rows"]:
        assert rust.sum_array_bool(np.array(x)) == exp
    for bad in (np.array([1, 1, 1, 1]), np.array([0.0, 0.0]), [True, True]):
        with pytest.raises(TypeError):
            rust.sum_array_bool(bad)
    for x, exp in g["sum_array_bool"]["rows"]:
        assert rust.sum_list_bool(x) == exp
    for bad in (np.array([1, 1]), np.array([0.0]), np.array([True, True]), [1, 1], [1.0, 0.0]):
        with pytest.raises(TypeError):
            rust.sum_list_bool(bad)
    for x, exp in g["sum_array"]["rows"]:
        assert rust.sum_array(np.array(x)) == exp
    for bad in (np.array([1, 1]), np.array([1, 0]), [True, True], np.array([True, False])):
        with pytest.raises(TypeError):
            rust.sum_array(bad)
    for x, exp in g["sum_list"]["rows"]:
        assert rust.sum_list(x) == exp
    assert rust.sum_list(np.array([True, True, False, False])) == 2
    assert np.array_equal(rust.list_to_zeros([1, 1, 1, 1]), np.zeros(4))
    for x, y, z in g["probify_float"]["rows"]:
        assert rust.probify_float(x, y, z) == np.clip(x, y, z)
    for x, s, t, exp in g["sigmoid"]["rows"]:
        assert round(rust.sigmoid(x, s, t), 4) == exp


def test_rust_shim_threshold_sigmoid_and_samplers():
    prm = {"impression_thresh": 0.05, "impression_bid_intercept": 0.4, "impression_slope": 6.0}
    halver = 2.0 + 1e-10
    th = min(max(halver * 0.05, 0), 1) / halver
    r = 1 / (1 + np.exp(-6.0 * (0.5 - 0.4)))
    assert rust.threshold_sigmoid(0.5, prm) == min(max((1 + 2 * th) * r - th, 0.0), 1.0)
    with pytest.raises(RuntimeError):
        rust.threshold_sigmoid(0.5, {"impression_thresh": 0.05})      # missing key panics in the reference
    rust.seed(7)
    a = [rust.nonneg_int_normal_sampler(16, 4) for _ in range(2000)]
    rust.seed(7)
    assert a == [rust.nonneg_int_normal_sampler(16, 4) for _ in range(2000)]
    assert abs(np.mean(a) - 16) < 0.4 and min(a) >= 0
    b = [rust.binomial_impressions(40, 0.25) for _ in range(2000)]
    assert abs(np.mean(b) - 10) < 0.3 and abs(np.var(b) - 7.5) < 1.0
    c = rust.cost_create(0.81, 20000)
    assert c.min() >= 0 and c.max() <= 4.4 + 1e-6 and round(c.mean(), 1) == 2.4     # 0.9/4 + 2.2
    with pytest.raises(RuntimeError):
        rust.binomial_impressions(3, 1.5)


def test_rust_shim_cost_model_moments():
    """the reference's own parity standard (adcraft/tests/rust/test_helpers.py:12-49): mean and std agree with
    generic_cost to 2 decimals over 100000 draws"""
    x = np.random.default_rng(40).uniform(0.01, 0.9, 100000)
    rng = np.random.default_rng(40)
    generic = np.around(np.clip(np.sqrt(x) / 4 + x / 2 + rng.normal(0, 1e-10 + np.sqrt(x) / 6), 0.0, x), 2)
    t = rust.cost_trans(x)
    assert round(t.mean(), 2) == round(generic.mean(), 2) and round(t.std(), 2) == round(generic.std(), 2)
    y = x.copy()
    rust.cost_mut(y)
    assert round(y.mean(), 2) == round(generic.mean(), 2) and round(y.std(), 2) == round(generic.std(), 2)


def test_repr_outcomes_format():
    o = [dict(bid=0.5, impressions=2, impression_share=0.5, buyside_clicks=1, costs=[0.25], sellside_conversions=1,
              revenues=[1.0], revenues_per_cost=[1.0], profit=0.75)]
    assert rust.repr_outcomes_py(o) == ("[{'bid': 0.5, 'impressions': 2, 'impression_share': 0.5, 'buyside_clicks': 1, "
                                        "'costs': [0.25], 'sellside_conversions': 1, 'revenues': [1.0], "
                                        "'revenues_per_cost': [1.0], 'profit': 0.75}]")


# ------------------------------------------------------------------ metrics + flatten (G5, G6)
def test_metrics_match_reference(golden):
    g = golden("g5_metrics.json")
    for c in g["akncp_ncp"]:
        kp, ip = np.array(c["kw_profits"]), np.array(c["ideal_profits"])
        assert float(em.compute_AKNCP(kp, ip)) == c["AKNCP"] and float(em.compute_NCP(kp, ip)) == c["NCP"]
    for c in g["max_expected"]:
        r = em.get_max_expected_bid_profits(c["kw_params"], np.array(c["cpc"]), np.array(c["ir"]))
        assert (float(r[0]), float(r[1]), int(r[2])) == (c["max_profit"], c["frac_positive"], c["argmax"])
    bids = np.array(g["bid_array"])
    for c in g["bid_curves"]:
        samples = (np.array(c["samples_cents"], dtype=np.float64) / 100.0).reshape(1, -1)

        class KW:
            def sample_bids(self, n):
                return samples
        ir, cpc = em.get_implicit_kw_bid_cpc_impressions(KW(), bids, n_samples=c["n_samples"])
        assert ir.tolist() == c["impression_rates"]
        np.testing.assert_allclose(cpc, c["cpc"], rtol=1e-12)


def test_flatten_matches_reference(golden):
    g = golden("g6_flatten.json")
    obs = {k: np.array(v) for k, v in g["obs"].items()}
    assert utils.flatten_dict_array(obs).tolist() == g["flat"]
    assert list(utils.FLAT_OBS_KEYS) == g["key_order"]


def test_header_is_plain_c_and_links(tmp_path):
    """the boundary is a C ABI: the header compiles as C99 and a C program links against the library"""
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(inc, "adcraft_engine.h")])
    exe = str(tmp_path / "c_abi_minimal")
    libdir = os.path.dirname(_ffi.library_path())
    _ffi.lib()
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", inc, os.path.join(ROOT, "examples", "c_abi_minimal.c"),
                           "-L", libdir, "-ladcraft_hip", "-Wl,-rpath," + libdir, "-o", exe])
    assert os.path.exists(exe)


def test_word_space_intervals_equal_auction_by_auction_resolution():
    """adc_law.h win_intervals (the product's threshold form of the 2nd-price clearing, run here on the host) against the
    oracle resolving the same words one auction at a time: sample the competitor's bid from the word, win iff bid > bid of
    the competitor (tie loses), the word 2^32 - 1 never wins.  Checked on the interval edges (every word where the outcome
    can change), the ends of the click / no-click sub-intervals, and random words - for the BASELINE keyword laws and for
    degenerate parameters."""
    import ctypes as C
    from oracle import capi as orc
    L, O = _ffi.lib(), orc.lib()
    rng = np.random.default_rng(12)
    cases = []
    for _ in range(60):
        loc = rng.uniform(0.3, 1.0)
        cases.append((round(rng.uniform(0.05, 1.6), 2), loc, max(0.01, rng.uniform(0.01, 0.3) * loc), rng.uniform(0.05, 0.95)))
    cases += [(0.01, 0.55, 0.08, 0.5), (0.02, 0.001, 0.004, 0.5), (5.0, 0.55, 0.08, 0.5), (1e7, 0.55, 0.08, 0.5), (0.7, 0.55, 0.08, 0.0),
              (0.7, 0.55, 0.08, 1.0), (0.7, 0.55, 0.0, 0.3), (0.7, 0.55, 1e-5, 0.3), (0.7, 0.55, 50.0, 0.3), (0.7, -0.4, 0.2, 0.3),
              (0.7, 0.55, -0.08, 0.3), (0.7, float("nan"), 0.08, 0.3), (0.7, 0.55, float("nan"), 0.3), (0.7, 0.55, float("inf"), 0.3),
              (0.7, 0.55, 0.08, 1e-9), (0.7, 0.55, 0.08, 1.0 - 1e-7), (0.64, 0.64, 0.05, 0.3), (0.3, 3.0, 0.2, 0.6)]
    out = (C.c_uint32 * 4)()
    click = C.c_int32()
    n_edges = 0
    for bid, loc, scale, ctr in cases:
        bid, loc, scale, ctr = (float(np.float32(x)) for x in (bid, loc, scale, ctr))
        assert L.adc_auction_word_intervals(bid, loc, scale, ctr, out) == 0
        c_lo, c_w, n_lo, n_w = (int(x) for x in out)
        bid_c = int(O.orc_bid_cents(bid))
        T = int(O.orc_bernoulli_threshold(ctr))
        words = {0, 1, 2**32 - 1, 2**32 - 2, max(T - 1, 0), min(T, 2**32 - 1), min(T + 1, 2**32 - 1)}
        for lo, w in ((c_lo, c_w), (n_lo, n_w)):
            for e in (lo - 1, lo, lo + 1, lo + w - 1, lo + w, lo + w + 1):
                if 0 <= e < 2**32:
                    words.add(e)
        n_edges += len(words)
        words |= set(int(x) for x in rng.integers(0, 2**32, 400, dtype=np.uint64))
        for w in words:
            comp = O.orc_auction_outcome(w, ctr, loc, scale, C.byref(click))
            win = bid_c > comp and w != 2**32 - 1
            cw = ((w - c_lo) & 0xFFFFFFFF) < c_w
            nw = ((w - n_lo) & 0xFFFFFFFF) < n_w
            assert not (cw and nw)
            assert (cw or nw) == win, (bid, loc, scale, ctr, w, comp)
            if win:
                assert cw == bool(click.value)
    assert n_edges > 500


def _bracket_keywords(rng, n):
    """keyword laws for the bracket checks: the BASELINE laws, wide log-uniform ranges, and degenerate corners"""
    third = n // 3
    bid = np.empty(n, np.float32); loc = np.empty(n, np.float32); scale = np.empty(n, np.float32); ctr = np.empty(n, np.float32)
    a = slice(0, third)
    bid[a] = rng.uniform(0.3, 1.0, third); loc[a] = rng.uniform(0.3, 1.0, third)
    scale[a] = np.maximum(0.01, rng.uniform(0.01, 0.3, third) * loc[a]); ctr[a] = rng.uniform(0.1, 0.9, third)
    b = slice(third, 2 * third)
    bid[b] = np.exp(rng.uniform(-5, 5, third)); loc[b] = np.exp(rng.uniform(-5, 5, third)) * rng.choice([-1.0, 1.0], third, p=[0.2, 0.8])
    scale[b] = np.exp(rng.uniform(-8, 4, third)); ctr[b] = np.where(rng.random(third) < 0.1, np.exp(-20 * rng.random(third)), rng.random(third))
    c = slice(2 * third, n)
    m = n - 2 * third
    bid[c] = np.rint(rng.uniform(1, 300, m)) / 100.0; loc[c] = rng.uniform(-0.5, 1.5, m); scale[c] = np.exp(rng.uniform(-14, 2, m))
    ctr[c] = rng.choice([0.0, 1.0, 1e-9, 1.0 - 1e-7, 0.5], m) * np.where(rng.random(m) < 0.5, 1.0, rng.random(m))
    for arr, vals in ((loc, [np.nan, np.inf, -np.inf, 0.0]), (scale, [np.nan, np.inf, 0.0, -0.08]), (bid, [np.nan, np.inf, -3.0, 1e7, 0.01])):
        idx = rng.integers(0, n, 4 * len(vals))
        arr[idx] = np.tile(np.array(vals, np.float32), 4)
    return bid, loc, scale, ctr


def test_win_brackets_enclose_the_exact_intervals():
    """adc_law.h win_brackets (what k_step_implicit_sparse classifies auctions with) against win_intervals (the exact word
    intervals, themselves checked word by word against the oracle above): inner <= exact <= outer for every keyword - BASELINE
    laws, log-uniform parameter ranges, zero / NaN / infinite / negative corners - and the in-between zone (the words that take
    the long way) stays a ~1e-5 share of the stream on the BASELINE laws."""
    import ctypes as C
    L = _ffi.lib()
    rng = np.random.default_rng(31)
    n = 600_000
    bid, loc, scale, ctr = _bracket_keywords(rng, n)
    first, amb = C.c_int64(-1), C.c_double(0.0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    bad = L.adc_check_win_brackets(n, ptr(bid), ptr(loc), ptr(scale), ptr(ctr), None, C.byref(first), C.byref(amb))
    assert bad == 0, (bad, first.value, bid[first.value], loc[first.value], scale[first.value], ctr[first.value])
    third = n // 3
    L.adc_check_win_brackets(third, ptr(bid), ptr(loc), ptr(scale), ptr(ctr), None, C.byref(first), C.byref(amb))
    assert amb.value / third / 2.0**32 < 1e-4
    # the single-keyword form returns what the batch checker saw; a degenerate competitor scale makes everything "in between"
    out = (C.c_uint32 * 8)()
    assert L.adc_auction_word_brackets(0.7, 0.55, 0.0, 0.3, out) == 0
    assert out[5] == 0 and out[7] == 0 and out[1] > 0
    assert L.adc_auction_word_brackets(0.7, 0.55, 0.08, 0.5, out) == 0
    x = (C.c_uint32 * 4)()
    assert L.adc_auction_word_intervals(0.7, 0.55, 0.08, 0.5, x) == 0
    assert out[0] <= x[0] <= out[4] and out[4] + out[5] <= x[0] + x[1] <= out[0] + out[1]
    assert 0 < (out[1] - out[5]) < 2**18 and out[5] > 2**28


def test_integration_doc_quotes_the_headers_abi_version():
    """INTEGRATION.md's ctypes stub asserts an ABI version: it must be the header's (round 4 shipped a stub that asserted 3 against a
    library of 4), and the versions the Python binding expects must be the header's too"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "adcraft_engine.h")).read()
    abi = int(re.search(r"#define ADC_ABI_VERSION (\d+)", header).group(1))
    rev = int(re.search(r"#define ADC_STREAM_REVISION (\d+)", header).group(1))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    quoted = [int(x) for x in re.findall(r"adc_abi_version\(\) == (\d+)", doc)]
    assert quoted and all(q == abi for q in quoted)
    assert (_ffi.ABI_VERSION, _ffi.STREAM_REVISION) == (abi, rev)
