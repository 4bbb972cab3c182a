"""adc_shims.cpp (+ adc_law.h) is plain host C++: compiled by g++ as a library of its own (oracle/build.py build_shims_host - the form
the sanitizers see, tools/run_sanitized.sh) it gives the same bits as the copy hipcc builds into libadcraft_hip.so."""
import ctypes as C
import os

import numpy as np

from adcraft_amd import _ffi
from oracle import build as obuild


def _host():
    L = C.CDLL(obuild.build_shims_host(sanitize=os.environ.get("ADCRAFT_ORACLE_SANITIZE") == "1"))
    f64, u64, i64, f32, vp = C.c_double, C.c_uint64, C.c_int64, C.c_float, C.c_void_p
    for name, (args, res) in {"adc_sigmoid": ([f64] * 3, f64), "adc_clamp": ([f64] * 3, f64), "adc_threshold_sigmoid": ([f64] * 4, f64),
                              "adc_sum_f64": ([vp, i64], f64), "adc_count_true": ([vp, i64], i64),
                              "adc_nonneg_int_normal": ([f64, f64, u64, u64], u64), "adc_binomial": ([u64, f64, u64, u64], u64),
                              "adc_cost_create": ([f64, i64, u64, u64, vp], C.c_int),
                              "adc_auction_word_intervals": ([f32] * 4 + [vp], C.c_int), "adc_auction_word_brackets": ([f32] * 4 + [vp], C.c_int),
                              "adc_check_win_brackets": ([i64] + [vp] * 7, i64)}.items():
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, res
    return L


def test_host_build_of_the_shims_equals_the_hipcc_build():
    H, P = _host(), _ffi.lib()
    rng = np.random.default_rng(5)
    for x, s, t in rng.normal(0, 2, (50, 3)):
        assert H.adc_sigmoid(x, s, t) == P.adc_sigmoid(x, s, t)
        assert H.adc_threshold_sigmoid(abs(x), 0.05, abs(s), 5 * abs(t)) == P.adc_threshold_sigmoid(abs(x), 0.05, abs(s), 5 * abs(t))
    v = rng.normal(0, 1, 1000)
    b = (rng.random(1000) < 0.3).astype(np.uint8)
    seq = 0.0
    for x in v:
        seq += float(x)                                  # rust.sum_list: left to right (src/lib.rs:310-312)
    assert H.adc_sum_f64(v.ctypes.data, v.size) == P.adc_sum_f64(v.ctypes.data, v.size) == seq
    assert H.adc_count_true(b.ctypes.data, b.size) == P.adc_count_true(b.ctypes.data, b.size) == int(b.sum())
    for i in range(200):
        m, sd, p = float(rng.uniform(0, 300)), float(rng.uniform(0, 60)), float(rng.random())
        assert H.adc_nonneg_int_normal(m, sd, 7, i) == P.adc_nonneg_int_normal(m, sd, 7, i)
        assert H.adc_binomial(i, p, 9, i) == P.adc_binomial(i, p, 9, i)
    a, c = np.zeros(37), np.zeros(37)
    assert H.adc_cost_create(0.8, 37, 3, 11, a.ctypes.data) == 0 and P.adc_cost_create(0.8, 37, 3, 11, c.ctypes.data) == 0
    assert np.array_equal(a, c) and a.min() >= 0.0 and a.max() <= 4.4
    n = 400
    bid = rng.uniform(0.01, 2.0, n).astype(np.float32)
    loc = rng.uniform(0.05, 1.2, n).astype(np.float32)
    scale = (loc * rng.uniform(0.01, 0.4, n)).astype(np.float32)
    ctr = rng.uniform(0.0, 1.0, n).astype(np.float32)
    for i in range(n):
        x4, y4, x8, y8 = (np.zeros(k, np.uint32) for k in (4, 4, 8, 8))
        H.adc_auction_word_intervals(bid[i], loc[i], scale[i], ctr[i], x4.ctypes.data)
        P.adc_auction_word_intervals(bid[i], loc[i], scale[i], ctr[i], y4.ctypes.data)
        assert np.array_equal(x4, y4)                    # exact intervals: table log, correctly rounded operations only
        H.adc_auction_word_brackets(bid[i], loc[i], scale[i], ctr[i], x8.ctypes.data)
        P.adc_auction_word_brackets(bid[i], loc[i], scale[i], ctr[i], y8.ctypes.data)
        assert np.array_equal(x8, y8)                    # (libm exp2f / a true division on both hosts)
    first, amb = C.c_int64(0), C.c_double(0.0)
    bad = H.adc_check_win_brackets(n, bid.ctypes.data, loc.ctypes.data, scale.ctypes.data, ctr.ctypes.data, None, C.byref(first), C.byref(amb))
    assert bad == 0 and first.value == -1 and amb.value / n < 2.0 ** 32 * 1e-3
