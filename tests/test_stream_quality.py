"""not gpu: the random stream on its production counter layout (oracle/stream_battery.c).

Philox4x32-7 is the paper's minimum Crush-resistant round count; what the kernels consume together are NEIGHBOURING
counters - the four words of a call, index j / j + 1, keyword k / k + 1, tick t / t + 1, env e / e + 1 - so those are the
pairs tested: serial correlation at lags 1..4, 2-D chi-squares on top and bottom bytes, uniformity, bit balance, per axis, on
>= 1e8 words; plus the independence of the (click, competitor uniform) pair the IMPLICIT path derives from one word."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import build as obuild

AXES = ["auction words in order", "index j, j+1", "keyword k, k+1", "tick t, t+1", "env e, e+1", "volume call k/4, k/4+1"]
STATS = ["corr lag 1", "corr lag 2", "corr lag 3", "corr lag 4", "chi2 top bytes", "chi2 bottom bytes", "top-byte uniformity", "popcount mean"]


def battery():
    L = C.CDLL(obuild.build_battery())
    L.bat_run.restype = C.c_double
    L.bat_run.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bat_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    return L


def run(rounds, calls_per_axis, seed):
    L = battery()
    stats = np.zeros((6, 8))
    pairs = np.zeros((4, 2))
    derived = np.zeros(3)
    words = L.bat_run(rounds, calls_per_axis, seed, stats.ctypes.data, pairs.ctypes.data, derived.ctypes.data)
    return words, stats, pairs, derived


def test_production_stream_passes_the_neighbouring_counter_battery():
    words, stats, pairs, derived = run(7, 5_000_000, 20261004)
    assert words >= 1.2e8                  # 6 axes x 2e7 words + 2e7 for the derived pair
    worst = np.abs(stats).max()
    assert worst < 5.0, [(AXES[a], STATS[s], float(stats[a, s])) for a, s in zip(*np.where(np.abs(stats) >= 5.0))]
    assert np.abs(pairs).max() < 5.0, pairs
    assert np.abs(derived).max() < 5.0, derived
    # the z-scores as a body: 48 + 8 + 3 statistics of a good generator have unit spread, not a shifted or inflated one
    allz = np.concatenate([stats.ravel(), pairs.ravel(), derived])
    assert abs(allz.mean()) < 0.6 and 0.6 < allz.std() < 1.5, (allz.mean(), allz.std())


def test_the_battery_sees_a_weak_generator():
    """the same battery on Philox with 2 and 3 rounds (known to fail) must fail loudly - it has teeth"""
    _, stats, pairs, derived = run(2, 400_000, 5)
    assert max(np.abs(stats).max(), np.abs(pairs).max()) > 50.0
    _, stats3, pairs3, _ = run(3, 1_000_000, 5)
    assert max(np.abs(stats3).max(), np.abs(pairs3).max()) > 8.0


def test_battery_philox_is_the_oracles_philox():
    from oracle import capi as orc
    L = battery()
    rng = np.random.default_rng(4)
    n = 1000
    ctr = rng.integers(0, 2**32, (n, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.integers(0, 2**32, (n, 2), dtype=np.uint64).astype(np.uint32)
    out = np.zeros((n, 4), np.uint32)
    for rounds in (7, 10):
        L.bat_philox(ctr.ctypes.data, key.ctypes.data, n, rounds, out.ctypes.data)
        for i in range(0, n, 97):
            assert [int(x) for x in out[i]] == [int(x) for x in orc.philox([int(c) for c in ctr[i]], [int(k) for k in key[i]], rounds)]
