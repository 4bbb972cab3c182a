"""Drop-in for the reference's pyo3 module `adcraft.rust` (src/lib.rs): same function names, argument
meaning and error behaviour, served by the C ABI of libadcraft_hip.so (include/adcraft_engine.h,
"scalar entry points").  `from adcraft_amd import rust` replaces `from adcraft import rust`.

Error conventions reproduced from the reference's own tests (adcraft/tests/rust/test_numpy_funcs.py):
wrong element types raise TypeError exactly where pyo3's extraction would.

The samplers (cost_create, binomial_impressions, nonneg_int_normal_sampler, cost_mut, cost_trans)
draw from an UNSEEDED thread_rng in the reference (src/lib.rs:25,43,61,75,320); here they draw from a
process-wide Philox counter stream that `seed(n)` can pin.
"""
import itertools
import os

import numpy as np

from . import _ffi

_counter = itertools.count(1)
_seed = int.from_bytes(os.urandom(8), "little")


def seed(n):
    """pin the sampler stream (the reference has no such call; its samplers are unseeded)"""
    global _seed, _counter
    _seed = int(n) & 0xFFFFFFFFFFFFFFFF
    _counter = itertools.count(1)


def _L():
    return _ffi.lib()


def _f64_array(x, what):
    if not (isinstance(x, np.ndarray) and x.dtype == np.float64):
        raise TypeError(f"{what}: argument must be a numpy ndarray of float64")
    return np.ascontiguousarray(x)


# ---- reducers (src/lib.rs:107-140) -----------------------------------------------------------------
def sum_array(x_vec):
    a = _f64_array(x_vec, "sum_array")
    return float(_L().adc_sum_f64(a.ctypes.data, a.size))


def sum_list(x_vec):
    """Vec<f64> extraction: sequences of int/float/bool and ndarrays are accepted (test_numpy_funcs.py:98-121)"""
    if isinstance(x_vec, (str, bytes)):
        raise TypeError("sum_list: Can't extract `str` to `Vec`")
    try:
        a = np.ascontiguousarray([float(v) for v in x_vec], dtype=np.float64)
    except (TypeError, ValueError) as e:
        raise TypeError(f"sum_list: {e}") from e
    return float(_L().adc_sum_f64(a.ctypes.data, a.size))


def sum_array_bool(x_vec):
    if not (isinstance(x_vec, np.ndarray) and x_vec.dtype == np.bool_):
        raise TypeError("sum_array_bool: argument must be a numpy ndarray of bool")
    a = np.ascontiguousarray(x_vec).view(np.uint8)
    return int(_L().adc_count_true(a.ctypes.data, a.size))


def sum_list_bool(x_vec):
    if isinstance(x_vec, np.ndarray) or not all(isinstance(v, (bool, np.bool_)) for v in x_vec):
        raise TypeError("sum_list_bool: argument must be a list of bool")
    a = np.ascontiguousarray(list(x_vec), dtype=np.uint8)
    return int(_L().adc_count_true(a.ctypes.data, a.size))


def array_to_zeros(x_vec):
    return np.zeros(_f64_array(x_vec, "array_to_zeros").size, dtype=np.float64)


def list_to_zeros(x_vec):
    return np.zeros(len([float(v) for v in x_vec]), dtype=np.float64)


# ---- deterministic scalars (src/lib.rs:78-105) ------------------------------------------------------
def sigmoid(x, s, t):
    return float(_L().adc_sigmoid(float(x), float(s), float(t)))


def probify_float(x, y, z):
    return float(_L().adc_clamp(float(x), float(y), float(z)))


def threshold_sigmoid(p, params):
    # get_value_with_default unwraps: a missing key is an error, the "defaults" never apply (src/lib.rs:302-308)
    try:
        th, ic, sl = params["impression_thresh"], params["impression_bid_intercept"], params["impression_slope"]
    except KeyError as e:
        raise RuntimeError(f"threshold_sigmoid: params is missing {e} (the reference panics here)") from e
    return float(_L().adc_threshold_sigmoid(float(p), float(th), float(ic), float(sl)))


# ---- samplers (src/lib.rs:17-76, 246-248, 314-325) --------------------------------------------------
def nonneg_int_normal_sampler(the_mean, std):
    if not std >= 0:
        raise RuntimeError("nonneg_int_normal_sampler: std must be >= 0 (Normal::new(...).unwrap() panics)")
    return int(_L().adc_nonneg_int_normal(float(the_mean), float(std), _seed, next(_counter)))


def binomial_impressions(n, p):
    if not (0.0 <= p <= 1.0):
        raise RuntimeError("binomial_impressions: p must be in [0, 1] (Binomial::new(...).unwrap() panics)")
    return int(_L().adc_binomial(int(n), float(p), _seed, next(_counter)))


def cost_create(x, n):
    out = np.zeros(int(n), dtype=np.float64)
    _ffi.check(_L().adc_cost_create(float(x), int(n), _seed, next(_counter), out.ctypes.data))
    return out


def _cost_law(p, z):
    sq = np.sqrt(p)
    return np.clip(sq / 4.0 + p / 2.0 + z * (1e-10 + sq / 6.0), 0.0, p)


def _normals(n):
    """n standard normals from the shim stream.  cost_create(x=36) is 3.7 + N(0, 1 + 1e-10) clamped to
    [0, 4.4], so it cannot serve; numpy's own counter-based Philox generator keyed by the shim seed does."""
    rng = np.random.Generator(np.random.Philox(key=_seed & (2**64 - 1), counter=next(_counter)))
    return rng.standard_normal(n)


def cost_trans(x_vec):
    """src/lib.rs:33-51: clamp(sqrt(p)/4 + p/2 + N(0,1)*(1e-10 + sqrt(p)/6), 0, p) - the *intended* cost model
    (test-only in the reference: adcraft/tests/rust/test_helpers.py:39-49)"""
    a = _f64_array(x_vec, "cost_trans")
    return _cost_law(a.ravel(), _normals(a.size))


def cost_mut(x_vec):
    """src/lib.rs:17-30: the same law, in place"""
    a = _f64_array(x_vec, "cost_mut")
    x_vec[...] = _cost_law(a, _normals(a.size).reshape(a.shape))


# ---- formatting (src/lib.rs:250-275) -----------------------------------------------------------------
def _fmt_f64(v):
    v = float(v)
    return repr(v) if not v.is_integer() else f"{v:.1f}"      # Rust {:?} of f64 prints 1.0, 0.25, ...


def _fmt_disp(v):
    v = float(v)
    return str(int(v)) if v.is_integer() and abs(v) < 1e16 else repr(v)     # Rust {} of f64 prints 1, 0.25


def repr_outcomes_py(outcomes):
    parts = []
    for o in outcomes:
        lst = lambda xs: "[" + ", ".join(_fmt_f64(x) for x in xs) + "]"   # noqa: E731
        parts.append("{" + f"'bid': {_fmt_disp(o['bid'])}, 'impressions': {int(o['impressions'])}, "
                     f"'impression_share': {_fmt_disp(o['impression_share'])}, 'buyside_clicks': {int(o['buyside_clicks'])}, "
                     f"'costs': {lst(o['costs'])}, 'sellside_conversions': {int(o['sellside_conversions'])}, "
                     f"'revenues': {lst(o['revenues'])}, 'revenues_per_cost': {lst(o['revenues_per_cost'])}, "
                     f"'profit': {_fmt_disp(o['profit'])}" + "}")
    if not parts:
        return "]"        # the reference pops two chars off "[" and pushes "]" (src/lib.rs:271-273)
    return "[" + ", ".join(parts) + "]"
