"""ctypes binding of include/adcraft_engine.h (the C ABI of the HIP engine).

No PyTorch, no fallbacks: if the shared library is missing and cannot be built, or if no HIP
device is usable, the error is raised to the caller.  ctypes releases the GIL during calls.
"""
import ctypes as C
import os

from . import build as _build

ADC_OK, ADC_EINVAL, ADC_EHIP, ADC_ENOMEM, ADC_ESTATE, ADC_ETYPE, ADC_ERCCL = 0, -1, -2, -3, -4, -5, -6
MODEL_IMPLICIT, MODEL_EXPLICIT, MODEL_IMPLICIT_GENERAL = 0, 1, 2
P_VOL_MEAN, P_VOL_STD, P_A, P_B, P_BCTR, P_SCTR, P_REV_MEAN, P_REV_STD, P_COUNT = range(9)
(BUF_PARAMS, BUF_BIDS, BUF_BUDGET, BUF_IMPRESSIONS, BUF_CLICKS, BUF_CONVERSIONS, BUF_COST, BUF_REVENUE, BUF_REWARD,
 BUF_CUM_PROFIT, BUF_DAYS, BUF_TERMINATED, BUF_TRUNCATED, BUF_METRIC_PROFIT, BUF_METRIC_SCALARS, BUF_FLAT_OBS) = range(16)


class EngineError(RuntimeError):
    pass


class EngineStateError(AssertionError):
    """ADC_ESTATE: the call is not valid in the engine's current state (step before reset, a step whose clicks can no longer be
    replayed ...).  An AssertionError, as the reference raises for step-before-reset (gymnasium_kw_env.py:194-196), but one of its
    own type, so that callers can tell it from a genuine Python `assert` failing."""


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device_id", C.c_int32), ("num_envs", C.c_int32),
                ("num_keywords", C.c_int32), ("model", C.c_int32), ("max_days", C.c_int32),
                ("loss_threshold", C.c_double), ("drift_vol", C.c_float), ("drift_ctr", C.c_float),
                ("drift_cvr", C.c_float), ("drift_enabled", C.c_int32), ("impression_thresh", C.c_float),
                ("auto_reset", C.c_int32), ("env_id_base", C.c_int64), ("seed", C.c_uint64)]


class StepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("impressions", "buyside_clicks", "sellside_conversions", "cost", "revenue",
                                          "reward", "cumulative_profit", "days_passed", "terminated", "truncated",
                                          "counts_u16", "counts_overflow")]


class Quantiles(C.Structure):
    _fields_ = [("buckets", C.c_int32 * 7), ("mins", C.c_void_p * 7), ("medians", C.c_void_p * 7), ("maxs", C.c_void_p * 7)]


class Tape(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("volumes", "bid_cents", "x_impressions", "x_cost", "click", "conv", "rev_cents")]
                + [(n, C.c_int64) for n in ("len_bid", "len_ximp", "len_xcost", "len_click", "len_conv", "len_rev")]
                + [(n, C.c_void_p) for n in ("off_bid", "off_ximp", "off_xcost", "off_click", "off_conv", "off_rev")]
                + [(n, C.c_void_p) for n in ("end_bid", "end_ximp", "end_xcost", "end_click", "end_conv", "end_rev")]
                + [("drift_uniforms", C.c_void_p)])


_lib = None
ABI_VERSION, STREAM_REVISION = 5, 5           # include/adcraft_engine.h ADC_ABI_VERSION, ADC_STREAM_REVISION


def library_path():
    return _build.LIB


def lib():
    """Load (building in-tree first if needed) libadcraft_hip.so; raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ADCRAFT_HIP_LIB") or _build.LIB      # (another build of this same library, for A/B timing)
    if path == _build.LIB and (not os.path.exists(path) or _build.stale()):
        try:
            _build.build()
        except Exception as exc:  # no hipcc / compile error: there is nothing to fall back to
            if not os.path.exists(path):
                raise EngineError(f"HIP engine library {path} is missing and could not be built: {exc}") from exc
            # an older build exists but the sources have changed since: running it would silently test stale kernels
            if os.environ.get("ADCRAFT_ALLOW_STALE_LIB") != "1":
                raise EngineError(f"HIP engine library {path} is older than its sources and the rebuild failed: {exc} "
                                  "(set ADCRAFT_ALLOW_STALE_LIB=1 to load the stale library anyway)") from exc
            import warnings
            warnings.warn(f"loading a STALE {path}: its sources changed and the rebuild failed ({exc})", RuntimeWarning)
    L = C.CDLL(path)
    L.adc_last_error.restype = C.c_char_p
    vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
    sig = {
        "adc_abi_version": ([], C.c_int),
        "adc_stream_revision": ([], C.c_int),
        "adc_device_count": ([vp], C.c_int),
        "adc_engine_create": ([C.POINTER(Config), C.POINTER(vp)], C.c_int),
        "adc_engine_destroy": ([vp], None),
        "adc_engine_bid_curves_build": ([vp, C.c_int, vp, C.c_int], C.c_int),
        "adc_engine_ideal_step": ([vp, vp, vp], C.c_int),
        "adc_engine_bid_curves_fetch": ([vp, vp, vp], C.c_int),
        "adc_engine_policy_oracle": ([vp, f32], C.c_int),
        "adc_engine_agent_init": ([vp, f32, vp], C.c_int),
        "adc_engine_agent_update": ([vp, vp, vp, vp], C.c_int),
        "adc_engine_agent_act": ([vp, f32, vp], C.c_int),
        "adc_engine_agent_step": ([vp, f32], C.c_int),
        "adc_engine_agent_state": ([vp, vp, vp, vp, vp, vp], C.c_int),
        "adc_engine_get_actions": ([vp, vp, vp], C.c_int),
        "adc_engine_metrics_read_nk": ([vp, vp, vp, vp], C.c_int),
        "adc_engine_run_days": ([vp, C.c_int, i32, f32], C.c_int),
        "adc_engine_day_graph_enable": ([vp, C.c_int], C.c_int),
        "adc_engine_set_params": ([vp, C.c_int, vp], C.c_int),
        "adc_engine_get_params": ([vp, C.c_int, vp], C.c_int),
        "adc_engine_set_env_params": ([vp, C.c_int, vp], C.c_int),
        "adc_engine_reset": ([vp, vp, vp], C.c_int),
        "adc_engine_generate_keywords": ([vp, C.POINTER(Quantiles), f32, C.c_uint32, vp], C.c_int),
        "adc_engine_generate_explicit_keywords": ([vp, C.c_uint32, vp], C.c_int),
        "adc_engine_set_limits": ([vp, i32, f64], C.c_int),
        "adc_engine_set_drift": ([vp, i32, f32, f32, f32], C.c_int),
        "adc_engine_get_rng_state": ([vp, vp, vp], C.c_int),
        "adc_engine_set_rng_state": ([vp, vp, vp], C.c_int),
        "adc_engine_get_episode_state": ([vp, vp, vp], C.c_int),
        "adc_engine_set_episode_state": ([vp, vp, vp], C.c_int),
        "adc_engine_step": ([vp, vp, vp, C.POINTER(StepOut)], C.c_int),
        "adc_engine_step_device": ([vp, vp, vp], C.c_int),
        "adc_engine_step_async": ([vp, vp, vp, C.POINTER(StepOut)], C.c_int),
        "adc_engine_step_flat_async": ([vp, vp, vp, vp, vp, vp], C.c_int),
        "adc_engine_wait": ([vp], C.c_int),
        "adc_engine_fetch": ([vp, C.POINTER(StepOut)], C.c_int),
        "adc_engine_out_offsets": ([vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], C.c_int),
        "adc_engine_synchronize": ([vp], C.c_int),
        "adc_engine_step_flat": ([vp, vp, vp, vp, vp, vp], C.c_int),
        "adc_engine_step_replay": ([vp, vp, vp, C.POINTER(Tape), C.POINTER(StepOut)], C.c_int),
        "adc_engine_update_keywords": ([vp], C.c_int),
        "adc_engine_set_general_model": ([vp, i32, f32, i32], C.c_int),
        "adc_host_alloc": ([C.c_size_t, C.POINTER(vp)], C.c_int),
        "adc_host_free": ([vp], None),
        "adc_engine_device_buffer": ([vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)], C.c_int),
        "adc_engine_stream": ([vp, C.POINTER(vp)], C.c_int),
        "adc_engine_sample_actions": ([vp, f32, f32, f32], C.c_int),
        "adc_engine_set_flat_actions_device": ([vp, vp], C.c_int),
        "adc_engine_flat_obs_enable": ([vp, C.c_int], C.c_int),
        "adc_engine_profile_enable": ([vp, C.c_int], C.c_int),
        "adc_engine_profile_sample_every": ([vp, C.c_int32], C.c_int),
        "adc_engine_profile_read": ([vp, vp, C.POINTER(i64)], C.c_int),
        "adc_engine_profile_records": ([vp, C.POINTER(i64)], C.c_int),
        "adc_engine_region_begin": ([vp], C.c_int),
        "adc_engine_region_end": ([vp, C.POINTER(f64)], C.c_int),
        "adc_engine_comm_stats": ([vp, C.POINTER(i64), C.POINTER(f64), C.POINTER(f64), C.c_int], C.c_int),
        "adc_engine_step_kernel_name": ([vp], C.c_char_p),
        "adc_engine_metrics_enable": ([vp, C.c_int], C.c_int),
        "adc_engine_metrics_reset": ([vp], C.c_int),
        "adc_engine_metrics_read": ([vp, vp, vp], C.c_int),
        "adc_engine_ideal_profit": ([vp, C.c_int, vp, C.c_int, vp], C.c_int),
        "adc_bid_curves_from_samples": ([C.c_int, vp, i32, vp, i32, vp, vp], C.c_int),
        "adc_engine_metrics_akncp_ncp": ([vp, f64, vp, vp], C.c_int),
        "adc_engine_bid_curves_contenders": ([vp, vp, vp, C.POINTER(i32)], C.c_int),
        "adc_engine_outcomes_replay": ([vp, i32, i32, vp, f32, i64, vp, vp, vp, vp, C.POINTER(i64), vp], C.c_int),
        "adc_engine_outcomes_replay_tape": ([vp, i32, vp, f32, C.POINTER(Tape), i64, vp, vp, vp, vp, C.POINTER(i64), vp], C.c_int),
        "adc_nth_price_auction": ([C.c_int, f64, vp, i32, i32, i32, i32, C.POINTER(i32), vp, vp], C.c_int),
        "adc_sigmoid": ([f64, f64, f64], f64),
        "adc_clamp": ([f64, f64, f64], f64),
        "adc_threshold_sigmoid": ([f64, f64, f64, f64], f64),
        "adc_sum_f64": ([vp, i64], f64),
        "adc_count_true": ([vp, i64], i64),
        "adc_nonneg_int_normal": ([f64, f64, u64, u64], u64),
        "adc_binomial": ([u64, f64, u64, u64], u64),
        "adc_cost_create": ([f64, i64, u64, u64, vp], C.c_int),
        "adc_auction_word_intervals": ([f32, f32, f32, f32, vp], C.c_int),
        "adc_auction_word_brackets": ([f32, f32, f32, f32, vp], C.c_int),
        "adc_check_win_brackets": ([i64, vp, vp, vp, vp, vp, vp, vp], i64),
        "adc_sample_random_keyword": ([C.c_uint64, C.c_uint32, C.c_uint32, vp], C.c_int),
        "adc_debug_win_brackets_device": ([C.c_int, i64, vp, vp, vp, vp, vp], C.c_int),
        "adc_debug_philox_device": ([C.c_int, i64, vp, vp, vp], C.c_int),
        "adc_debug_walk_stats": ([vp, vp, C.c_int], C.c_int),
        "adc_debug_direct_days": ([vp, vp, C.c_int], C.c_int),
        "adc_debug_chain_device": ([C.c_int, f64, i64, vp, vp], C.c_int),
        "adc_engine_env_groups": ([vp, vp], C.c_int),
        "adc_engine_set_env_groups": ([vp, C.c_int32], C.c_int),
        "adc_comm_get_unique_id": ([vp], C.c_int),
        "adc_engine_comm_init": ([vp, vp, i32, i32], C.c_int),
        "adc_engine_comm_destroy": ([vp], C.c_int),
        "adc_engine_comm_info": ([vp, vp, vp], C.c_int),
        "adc_engine_metrics_allreduce": ([vp, vp, vp, vp], C.c_int),
        "adc_engine_comm_allreduce_f64": ([vp, vp, i32, i32], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)      # AttributeError here = the .so does not export what the header declares
        fn.argtypes = args
        fn.restype = res
    if L.adc_abi_version() != ABI_VERSION or L.adc_stream_revision() != STREAM_REVISION:
        raise EngineError(f"{path}: ABI version {L.adc_abi_version()} / stream revision {L.adc_stream_revision()}, this package "
                          f"expects {ABI_VERSION} / {STREAM_REVISION} (a stale build?)")
    _lib = L
    return L


EXPORTED = None  # filled by tests from include/adcraft_engine.h


def check(rc):
    """Map adc_status to the exceptions the reference's callers see (SURVEY 8b error conventions)."""
    if rc == ADC_OK:
        return
    msg = (lib().adc_last_error() or b"").decode("utf-8", "replace")
    if rc == ADC_EINVAL:
        raise ValueError(msg)
    if rc == ADC_ESTATE:
        raise EngineStateError(msg)        # (an AssertionError: gymnasium_kw_env.py:194-196 asserts on step-before-reset)
    if rc == ADC_ENOMEM:
        raise MemoryError(msg)
    if rc == ADC_ETYPE:
        raise TypeError(msg)
    raise EngineError(msg)


def ptr(a):
    return None if a is None else a.ctypes.data


def device_count():
    n = C.c_int(0)
    rc = lib().adc_device_count(C.byref(n))
    return n.value if rc == ADC_OK else 0
