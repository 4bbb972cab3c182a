"""StepEngine: thin object wrapper over the C ABI (include/adcraft_engine.h).

One engine = N environments x K keywords resident on one MI355X.  The per-step hot path of
the reference (adcraft/gymnasium_kw_env.py:160-269 -> adcraft/bidding_simulation.py:170-234)
is ONE call here for all environments.  numpy in / numpy out; device-resident variants for
consumers that keep actions and observations in HBM.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import (MODEL_EXPLICIT, MODEL_IMPLICIT, P_A, P_B, P_BCTR, P_COUNT, P_REV_MEAN, P_REV_STD, P_SCTR,  # noqa: F401
                   P_VOL_MEAN, P_VOL_STD, check, ptr)

_OUT_SPEC = (("impressions", np.int32, True), ("buyside_clicks", np.int32, True),
             ("sellside_conversions", np.int32, True), ("cost", np.float32, True), ("revenue", np.float32, True),
             ("reward", np.float64, False), ("cumulative_profit", np.float64, False), ("days_passed", np.int32, False),
             ("terminated", np.uint8, False), ("truncated", np.uint8, False))


_COUNT_NAMES = ("impressions", "buyside_clicks", "sellside_conversions")


def _out_offsets(lib, handle):
    off, total = (C.c_size_t * 10)(), C.c_size_t()
    check(lib.adc_engine_out_offsets(handle, off, C.byref(total)))
    return list(off), total.value


def _views_at(block, offsets, N, K, names):
    """numpy views of a byte block at the engine's output offsets"""
    out = {}
    for (name, dt, per_kw), off in zip(_OUT_SPEC, offsets):
        if name in names:
            count = N * K if per_kw else N
            out[name] = block[off:off + count * np.dtype(dt).itemsize].view(dt).reshape((N, K) if per_kw else (N,))
    return out


def _alloc_outputs(alloc, N, K, compact, offsets, block_bytes):
    """the step outputs of one engine as views of ONE page-locked block laid out like the engine's device block, so that they
    come back in one transfer (adc_engine_out_offsets).  With compact counts the three count arrays are uint16 views [N, K] of
    a [N, 3, K] array the device packs (adc_step_out.counts_u16) and no int32 counts are transferred."""
    names = [n for n, _, _ in _OUT_SPEC if not (compact and n in _COUNT_NAMES)]
    out = _views_at(alloc((block_bytes,), np.uint8), offsets, N, K, names)
    if not compact:
        return out, None, None
    packed, overflow = alloc((N, 3, K), np.uint16), alloc((1,), np.int32)
    for i, name in enumerate(_COUNT_NAMES):
        out[name] = packed[:, i, :]
    return out, packed, overflow


def _out_struct(out, packed, overflow, b0, b1, small=None):
    """adc_step_out over rows b0:b1 of the output arrays (small: this part's own per-env arrays)"""
    src = lambda name, per_kw: (out[name][b0:b1] if per_kw or small is None else small[name]).ctypes.data   # noqa: E731
    if packed is None:
        return _ffi.StepOut(*(src(name, per_kw) for name, _, per_kw in _OUT_SPEC), None, None)
    ptrs = [None if name in _COUNT_NAMES else src(name, per_kw) for name, _, per_kw in _OUT_SPEC]
    return _ffi.StepOut(*ptrs, packed[b0:b1].ctypes.data, overflow[b0:b0 + 1].ctypes.data if overflow.size > 1 else overflow.ctypes.data)


def _is_buffer(a, buf):
    """a is the page-locked array buf itself (or a full view of it): nothing to stage"""
    return isinstance(a, np.ndarray) and a.dtype == buf.dtype and a.size == buf.size and a.flags.c_contiguous \
        and a.__array_interface__["data"][0] == buf.__array_interface__["data"][0]


def _check_overflow(overflow):
    if overflow is not None and overflow.any():
        overflow[...] = 0
        raise OverflowError("a keyword count exceeded 65535: construct the engine / env with compact_counts=False")


class StepEngine:
    def __init__(self, num_envs, num_keywords, model=MODEL_IMPLICIT, *, device_id=0, max_days=60,
                 loss_threshold=10000.0, drift=(0.03, 0.03, 0.03), drift_enabled=False, impression_thresh=0.05,
                 auto_reset=False, env_id_base=0, seed=0, compact_counts=False):
        self._h = None
        self._lib = _ffi.lib()
        self.num_envs, self.num_keywords, self.model = int(num_envs), int(num_keywords), int(model)
        self.max_days = int(max_days)
        cfg = _ffi.Config(C.sizeof(_ffi.Config), int(device_id), self.num_envs, self.num_keywords, self.model,
                          int(max_days), float(loss_threshold), float(drift[0]), float(drift[1]), float(drift[2]),
                          1 if drift_enabled else 0, float(impression_thresh), 1 if auto_reset else 0,
                          int(env_id_base), int(seed) & 0xFFFFFFFFFFFFFFFF)
        h = C.c_void_p()
        check(self._lib.adc_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        N, K = self.num_envs, self.num_keywords
        # step I/O buffers live in page-locked host memory: observations DMA straight into these numpy arrays
        self._pinned = []
        self.compact_counts = bool(compact_counts)
        self.out, self._counts_u16, self._overflow = _alloc_outputs(self._pinned_array, N, K, self.compact_counts,
                                                                    *_out_offsets(self._lib, self._h))
        self._out = _out_struct(self.out, self._counts_u16, self._overflow, 0, N)
        self._flat_io = None
        self._bids_stage = self._pinned_array((N, K), np.float32)
        self._budget_stage = self._pinned_array((N,), np.float32)

    def _pinned_array(self, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(self._lib.adc_host_alloc(max(nbytes, 1), C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        a[...] = 0
        return a

    # ---- lifecycle
    def close(self):
        if self._h is not None:
            self._lib.adc_engine_destroy(self._h)
            self._h = None
            self.out, self._bids_stage, self._budget_stage, self._flat_io = {}, None, None, None      # drop views before freeing
            self._counts_u16, self._overflow = None, None
            for p in self._pinned:
                self._lib.adc_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- keyword state
    def set_params(self, param_id, values):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float32),
                                                 (self.num_envs, self.num_keywords)))
        check(self._lib.adc_engine_set_params(self._h, int(param_id), a.ctypes.data))

    def set_all_params(self, planes):
        """planes: float array [8][N][K] (or broadcastable to it)"""
        planes = np.asarray(planes, dtype=np.float32)
        for p in range(P_COUNT):
            self.set_params(p, planes[p])

    def set_env_params(self, env, planes_8k):
        a = np.ascontiguousarray(planes_8k, dtype=np.float32).reshape(P_COUNT, self.num_keywords)
        check(self._lib.adc_engine_set_env_params(self._h, int(env), a.ctypes.data))

    def get_params(self, param_id):
        a = np.zeros((self.num_envs, self.num_keywords), dtype=np.float32)
        check(self._lib.adc_engine_get_params(self._h, int(param_id), a.ctypes.data))
        return a

    def get_all_params(self):
        return np.stack([self.get_params(p) for p in range(P_COUNT)])

    def reset(self, env_mask=None, seeds=None):
        m = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8)
        s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        if m is not None and m.shape != (self.num_envs,):
            raise ValueError("env_mask must have shape (num_envs,)")
        if s is not None and s.shape != (self.num_envs,):
            raise ValueError("seeds must have shape (num_envs,)")
        check(self._lib.adc_engine_reset(self._h, ptr(m), ptr(s)))

    QUANTITIES = ("vol", "ave_cpc", "std_cpc", "bctr", "sctr", "rpsc", "std_rpsc")

    def generate_keywords(self, table, no_vol_prob=0.0, env_mask=None, serial=0):
        """draw every (masked) env's keyword set on the device from a quantile table (dict of columns
        count_/min_/median_/max_<quantity>, as the reference's DataFrame); see adc_engine_generate_keywords"""
        q = _ffi.Quantiles()
        keep = []
        for i, name in enumerate(self.QUANTITIES):
            col = lambda c: np.asarray(table[f"{c}_{name}"].to_numpy() if hasattr(table[f"{c}_{name}"], "to_numpy")  # noqa: E731
                                       else table[f"{c}_{name}"], dtype=np.float64)
            sel = col("count") > 0 if f"count_{name}" in table and name != "vol" else np.ones(len(col("min")), bool)
            arrs = [np.ascontiguousarray(col(c)[sel], dtype=np.float32) for c in ("min", "median", "max")]
            keep.append(arrs)
            q.buckets[i] = arrs[0].size
            q.mins[i], q.medians[i], q.maxs[i] = (a.ctypes.data for a in arrs)
        m = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8)
        check(self._lib.adc_engine_generate_keywords(self._h, C.byref(q), float(no_vol_prob), int(serial), ptr(m)))

    def generate_explicit_keywords(self, env_mask=None, serial=0):
        """draw every (masked) env's EXPLICIT keyword set on the device: the law of sample_random_keywords
        (gymnasium_kw_utils.py:113-156), from each env's own Philox key; see adc_engine_generate_explicit_keywords"""
        m = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8)
        check(self._lib.adc_engine_generate_explicit_keywords(self._h, int(serial), ptr(m)))

    def set_limits(self, max_days, loss_threshold):
        check(self._lib.adc_engine_set_limits(self._h, int(max_days), float(loss_threshold)))
        self.max_days = int(max_days)

    def set_general_model(self, max_bidders=30, participation_rate=0.6, num_winners=1):
        """model=2 (the reference's default ImplicitKeyword): bidder pool and number of winning placements"""
        check(self._lib.adc_engine_set_general_model(self._h, int(max_bidders), float(participation_rate), int(num_winners)))

    def set_drift(self, enabled, drift=(0.03, 0.03, 0.03)):
        check(self._lib.adc_engine_set_drift(self._h, 1 if enabled else 0, float(drift[0]), float(drift[1]), float(drift[2])))

    def get_rng_state(self):
        k = np.zeros(self.num_envs, dtype=np.uint64)
        t = np.zeros(self.num_envs, dtype=np.uint32)
        check(self._lib.adc_engine_get_rng_state(self._h, k.ctypes.data, t.ctypes.data))
        return k, t

    def set_rng_state(self, keys=None, ticks=None):
        k = None if keys is None else np.ascontiguousarray(keys, dtype=np.uint64)
        t = None if ticks is None else np.ascontiguousarray(ticks, dtype=np.uint32)
        check(self._lib.adc_engine_set_rng_state(self._h, ptr(k), ptr(t)))

    def get_episode_state(self):
        d = np.zeros(self.num_envs, dtype=np.int32)
        c = np.zeros(self.num_envs, dtype=np.float64)
        check(self._lib.adc_engine_get_episode_state(self._h, d.ctypes.data, c.ctypes.data))
        return d, c

    def set_episode_state(self, day=None, cum_profit=None):
        d = None if day is None else np.ascontiguousarray(day, dtype=np.int32)
        c = None if cum_profit is None else np.ascontiguousarray(cum_profit, dtype=np.float64)
        check(self._lib.adc_engine_set_episode_state(self._h, ptr(d), ptr(c)))

    # ---- the hot path
    def _actions(self, bids, budget):
        if not _is_buffer(bids, self._bids_stage):
            self._bids_stage[...] = np.asarray(bids, dtype=np.float32).reshape(-1, self.num_keywords) if np.ndim(bids) else bids
        if not _is_buffer(budget, self._budget_stage):
            self._budget_stage[...] = budget
        return self._bids_stage, self._budget_stage

    def action_buffers(self):
        """(bids [N, K], budget [N]) page-locked arrays: fill them in place and pass them to step() to skip the staging copy"""
        return self._bids_stage, self._budget_stage

    def step(self, bids, budget, copy=True):
        """host in / host out, synchronous.  Returns dict of numpy arrays (views of reused buffers if copy=False)."""
        b, g = self._actions(bids, budget)
        check(self._lib.adc_engine_step(self._h, b.ctypes.data, g.ctypes.data, C.byref(self._out)))
        _check_overflow(self._overflow)
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out

    def outcomes_replay(self, env, bids_k, budget, steps_back=1, tape=None):
        """the paid clicks of one step of env `env` (steps_back = 1: its last), one by one, in the reference's order
        (adc_engine_outcomes_replay): dict of keyword, timestep, cost (dollars), revenue (dollars, -1 = no conversion) and
        share_volume [K].  An earlier step can be replayed while drift is off and nothing has changed the parameters since.
        With a ReplayTape the step is the one the tape describes (adc_engine_outcomes_replay_tape: parity against the reference's
        recorded BiddingOutcomes); nothing of the engine's state is touched either way."""
        K = self.num_keywords
        bids = np.ascontiguousarray(bids_k, dtype=np.float32).reshape(K)
        share = np.zeros(K, np.int32)
        n = C.c_int64(0)
        cap = 4096
        while True:
            kw, ts = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            cost, rev = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
            if tape is None:
                check(self._lib.adc_engine_outcomes_replay(self._h, int(env), int(steps_back), bids.ctypes.data, float(budget), cap, kw.ctypes.data,
                                                           ts.ctypes.data, cost.ctypes.data, rev.ctypes.data, C.byref(n), share.ctypes.data))
            else:
                check(self._lib.adc_engine_outcomes_replay_tape(self._h, int(env), bids.ctypes.data, float(budget), C.byref(tape.struct), cap,
                                                                kw.ctypes.data, ts.ctypes.data, cost.ctypes.data, rev.ctypes.data, C.byref(n),
                                                                share.ctypes.data))
            if n.value <= cap:
                m = n.value
                return dict(keyword=kw[:m], timestep=ts[:m], cost=cost[:m], revenue=rev[:m], share_volume=share)
            cap = int(n.value)

    def step_flat(self, flat_actions):
        """FlatArrayWrapper-layout step: actions float32 [N, K+1] = [budget, bids...] -> (flat_obs [N, 5K+2], reward,
        terminated, truncated); the returned arrays are views of page-locked buffers, valid until the next step"""
        N, K = self.num_envs, self.num_keywords
        if self._flat_io is None:
            self._flat_io = (self._pinned_array((N, K + 1), np.float32), self._pinned_array((N, 5 * K + 2), np.float32))
        act, obs = self._flat_io
        act[...] = flat_actions
        check(self._lib.adc_engine_step_flat(self._h, act.ctypes.data, obs.ctypes.data, self.out["reward"].ctypes.data,
                                             self.out["terminated"].ctypes.data, self.out["truncated"].ctypes.data))
        return obs, self.out["reward"], self.out["terminated"], self.out["truncated"]

    def step_async(self, bids_ptr, budget_ptr, out_struct=None):
        """enqueue a host-in / host-out step and return (adc_engine_step_async); the buffers (page-locked) must stay
        untouched until wait()"""
        check(self._lib.adc_engine_step_async(self._h, bids_ptr, budget_ptr, C.byref(self._out if out_struct is None else out_struct)))

    def step_flat_async(self, act_ptr, obs_ptr, reward_ptr, term_ptr, trunc_ptr):
        check(self._lib.adc_engine_step_flat_async(self._h, act_ptr, obs_ptr, reward_ptr, term_ptr, trunc_ptr))

    def wait(self):
        check(self._lib.adc_engine_wait(self._h))

    def step_replay(self, bids, budget, tape, copy=True):
        b, g = self._actions(bids, budget)
        check(self._lib.adc_engine_step_replay(self._h, b.ctypes.data, g.ctypes.data, C.byref(tape.struct), C.byref(self._out)))
        _check_overflow(self._overflow)
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out

    def step_device(self, d_bids=None, d_budget=None):
        """asynchronous; None = the engine's staging buffers (see sample_actions / device_buffer)."""
        check(self._lib.adc_engine_step_device(self._h, d_bids, d_budget))

    def fetch(self, copy=True):
        check(self._lib.adc_engine_fetch(self._h, C.byref(self._out)))
        _check_overflow(self._overflow)
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out

    def synchronize(self):
        check(self._lib.adc_engine_synchronize(self._h))

    def update_keywords(self):
        check(self._lib.adc_engine_update_keywords(self._h))

    def sample_actions(self, bid_lo=0.30, bid_hi=1.00, budget=1.0e9):
        check(self._lib.adc_engine_sample_actions(self._h, bid_lo, bid_hi, budget))

    def flat_obs_enable(self, on=True):
        check(self._lib.adc_engine_flat_obs_enable(self._h, 1 if on else 0))

    def set_flat_actions_device(self, d_flat_ptr):
        """device pointer to float32 [N][K+1] = [budget, bids...] -> the engine's staging buffers (async)"""
        check(self._lib.adc_engine_set_flat_actions_device(self._h, d_flat_ptr))

    def device_buffer(self, buffer_id):
        p = C.c_void_p()
        n = C.c_size_t()
        check(self._lib.adc_engine_device_buffer(self._h, int(buffer_id), C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream(self):
        p = C.c_void_p()
        check(self._lib.adc_engine_stream(self._h, C.byref(p)))
        return p.value

    # ---- measurement / metrics
    def profile_enable(self, on=True, every=1):
        """HIP events around the kernels of every `every`-th step (recording them costs ~16 us a step: sample when timing)"""
        check(self._lib.adc_engine_profile_sample_every(self._h, int(every)))
        check(self._lib.adc_engine_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """(ms_fast_pass, ms_exact_pass_and_tail, ms_metric_accumulate), measured steps - summed since the last read"""
        ms = (C.c_double * 3)()
        n = C.c_int64()
        check(self._lib.adc_engine_profile_read(self._h, ms, C.byref(n)))
        return tuple(ms), n.value

    def profile_records(self):
        """event records issued since the engine was created (four per bracketed step; none while profiling is off)"""
        n = C.c_int64()
        check(self._lib.adc_engine_profile_records(self._h, C.byref(n)))
        return n.value

    def walk_stats(self, reset=False):
        """k_step_click_walk's counters on this device: [walked, list overflowed, campaign stopped, other hand-overs] (adc_debug_walk_stats)"""
        out = np.zeros(4, dtype=np.int64)
        check(self._lib.adc_debug_walk_stats(self._h, out.ctypes.data, 1 if reset else 0))
        return out

    def direct_days(self, reset=False):
        """env-days handed to k_step_rest_of_day without the row kernel, on this device (adc_debug_direct_days)"""
        out = np.zeros(1, dtype=np.int64)
        check(self._lib.adc_debug_direct_days(self._h, out.ctypes.data, 1 if reset else 0))
        return int(out[0])

    def env_groups(self):
        """how many env groups (streams) the last IMPLICIT step ran as (adc_engine_env_groups); scheduling only"""
        n = C.c_int32(1)
        check(self._lib.adc_engine_env_groups(self._h, C.byref(n)))
        return n.value

    def set_env_groups(self, groups):
        """0: the engine chooses how many env groups a step runs as; 1..4: that many (adc_engine_set_env_groups)"""
        check(self._lib.adc_engine_set_env_groups(self._h, int(groups)))

    def step_kernel_name(self):
        """the first-pass kernel of the last step (the one profile_read()'s first duration times)"""
        return self._lib.adc_engine_step_kernel_name(self._h).decode()

    def metrics_enable(self, on=True):
        check(self._lib.adc_engine_metrics_enable(self._h, 1 if on else 0))

    def metrics_reset(self):
        check(self._lib.adc_engine_metrics_reset(self._h))

    def ideal_profit(self, n_samples=2048, bid_grid=None):
        """max expected profit per keyword from the current parameters (experiment_metrics.py:20-61), dollars [N, K];
        bid_grid defaults to the notebooks' np.arange(0.01, 3.00, 0.01)"""
        grid = np.ascontiguousarray(np.arange(0.01, 3.00, 0.01) if bid_grid is None else bid_grid, dtype=np.float64)
        out = np.zeros((self.num_envs, self.num_keywords), dtype=np.float64)
        check(self._lib.adc_engine_ideal_profit(self._h, int(n_samples), grid.ctypes.data, grid.size, out.ctypes.data))
        return out

    def metrics_read(self):
        kp = np.zeros(self.num_keywords, dtype=np.int64)
        sc = np.zeros(8, dtype=np.int64)
        check(self._lib.adc_engine_metrics_read(self._h, kp.ctypes.data, sc.ctypes.data))
        return kp, sc


    # ---- device-resident callers of the step: per-step ideal profit and the baseline bidders --------------------
    def bid_curves_build(self, n_samples=2048, bid_grid=None):
        """cache get_implicit_kw_bid_cpc_impressions of every keyword on the device (experiment_metrics.py:20-37)"""
        grid = np.ascontiguousarray(np.arange(0.01, 3.00, 0.01) if bid_grid is None else bid_grid, dtype=np.float64)
        check(self._lib.adc_engine_bid_curves_build(self._h, int(n_samples), grid.ctypes.data, grid.size))
        self._bid_grid = grid

    def bid_curves_fetch(self):
        """the cached curves as host arrays (impression_rate, cpc), each [N, K, n_bids]"""
        nb = self._bid_grid.size
        ir = np.zeros((self.num_envs, self.num_keywords, nb), np.float64)
        cpc = np.zeros((self.num_envs, self.num_keywords, nb), np.float64)
        check(self._lib.adc_engine_bid_curves_fetch(self._h, ir.ctypes.data, cpc.ctypes.data))
        return ir, cpc

    def bid_curves_contenders(self):
        """(count [N, K] (65535 = the whole grid), grid indices [N, K, cap], margin intervals [N, K, cap, 2]) of the curve points
        the per-step ideal chooses from"""
        cap = C.c_int32(0)
        check(self._lib.adc_engine_bid_curves_contenders(self._h, None, None, C.byref(cap)))
        n = np.zeros((self.num_envs, self.num_keywords), np.uint16)
        ent = np.zeros((self.num_envs, self.num_keywords, cap.value, 6), np.uint32)
        check(self._lib.adc_engine_bid_curves_contenders(self._h, n.ctypes.data, ent.ctypes.data, C.byref(cap)))
        return n, ent[..., 4].astype(np.int32), ent[..., 0:2].copy().view(np.float32)

    def ideal_step(self, fetch=True):
        """get_max_expected_bid_profits for the current parameters against the cached curves; with metrics enabled
        the ideal is also accumulated.  Returns (ideal [N, K] dollars, argmax index [N, K]) or None if not fetch."""
        if not fetch:
            check(self._lib.adc_engine_ideal_step(self._h, None, None))
            return None
        ideal = np.zeros((self.num_envs, self.num_keywords), dtype=np.float64)
        best = np.zeros((self.num_envs, self.num_keywords), dtype=np.int32)
        check(self._lib.adc_engine_ideal_step(self._h, ideal.ctypes.data, best.ctypes.data))
        return ideal, best

    def policy_oracle(self, budget=100000.0):
        """next action := the grid bid of maximum expected profit (after ideal_step)"""
        check(self._lib.adc_engine_policy_oracle(self._h, float(budget)))

    def agent_init(self, default_rpc=3.0, seeds=None):
        sd = None if seeds is None else np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.num_envs,)))
        check(self._lib.adc_engine_agent_init(self._h, float(default_rpc), None if sd is None else sd.ctypes.data))

    def agent_update(self, clicks=None, conversions=None, revenue=None):
        if clicks is None and conversions is None and revenue is None:
            check(self._lib.adc_engine_agent_update(self._h, None, None, None))
            return
        shape = (self.num_envs, self.num_keywords)
        c = np.ascontiguousarray(np.asarray(clicks).reshape(shape), dtype=np.int32)
        v = np.ascontiguousarray(np.asarray(conversions).reshape(shape), dtype=np.int32)
        r = np.ascontiguousarray(np.asarray(revenue).reshape(shape), dtype=np.float32)
        check(self._lib.adc_engine_agent_update(self._h, c.ctypes.data, v.ctypes.data, r.ctypes.data))

    def agent_act(self, budget_override=0.0, replay_uniforms=None):
        u = None
        if replay_uniforms is not None:
            u = np.ascontiguousarray(np.asarray(replay_uniforms, dtype=np.float64).reshape(self.num_envs, self.num_keywords))
        check(self._lib.adc_engine_agent_act(self._h, float(budget_override), None if u is None else u.ctypes.data))

    def agent_step(self, budget_override=0.0):
        check(self._lib.adc_engine_agent_step(self._h, float(budget_override)))

    def agent_state(self):
        shape = (self.num_envs, self.num_keywords)
        st = dict(ave_rpc=np.zeros(shape, np.float32), num_rpc_obs=np.zeros(shape, np.int32), ave_sctr=np.zeros(shape, np.float32),
                  num_sctr_obs=np.zeros(shape, np.int32), max_bids=np.zeros(shape, np.float64))
        check(self._lib.adc_engine_agent_state(self._h, *(st[k].ctypes.data for k in ("ave_rpc", "num_rpc_obs", "ave_sctr",
                                                                                      "num_sctr_obs", "max_bids"))))
        return st

    def get_actions(self):
        bids = np.zeros((self.num_envs, self.num_keywords), np.float32)
        budget = np.zeros(self.num_envs, np.float32)
        check(self._lib.adc_engine_get_actions(self._h, bids.ctypes.data, budget.ctypes.data))
        return bids, budget

    POLICIES = {"fixed": 0, "zero_margin": 1, "oracle": 2}

    # ---- multi-GPU: the episode-metric all-reduce (RCCL behind the C ABI; adcraft_amd/comm.py brings it up) ----------
    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        check(self._lib.adc_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world_size):
        uid = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        check(self._lib.adc_engine_comm_init(self._h, uid, int(rank), int(world_size)))

    def comm_destroy(self):
        check(self._lib.adc_engine_comm_destroy(self._h))

    def comm_info(self):
        r, w = C.c_int32(0), C.c_int32(1)
        check(self._lib.adc_engine_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_stats(self, reset=False):
        """(calls, ms of this rank's own reduction kernels, ms of the ncclAllReduce) of the metric reductions so far"""
        n, a, b = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        check(self._lib.adc_engine_comm_stats(self._h, C.byref(n), C.byref(a), C.byref(b), 1 if reset else 0))
        return n.value, a.value, b.value

    def region_begin(self):
        """one event on the engine's stream; region_end() -> GPU milliseconds since (one event pair for a whole timed region)"""
        check(self._lib.adc_engine_region_begin(self._h))

    def region_end(self):
        ms = C.c_double(0.0)
        check(self._lib.adc_engine_region_end(self._h, C.byref(ms)))
        return ms.value

    def metrics_allreduce(self, ideal_k=None, ideal_pos_k=None):
        """the one collective of the path: (profit_cents[K], ideal[K], ideal_pos[K], scalars[8]) summed over steps, envs and
        the ranks of the engine's communicator (this rank alone without one)"""
        K = self.num_keywords
        out = np.zeros(3 * K + 8, dtype=np.float64)
        a = None if ideal_k is None else np.ascontiguousarray(ideal_k, dtype=np.float64)
        b = None if ideal_pos_k is None else np.ascontiguousarray(ideal_pos_k, dtype=np.float64)
        check(self._lib.adc_engine_metrics_allreduce(self._h, None if a is None else a.ctypes.data,
                                                     None if b is None else b.ctypes.data, out.ctypes.data))
        return out[:K], out[K:2 * K], out[2 * K:3 * K], out[3 * K:]

    def comm_allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(self._lib.adc_engine_comm_allreduce_f64(self._h, v.ctypes.data, v.size, 1 if op == "max" else 0))
        return v

    def run_days(self, policy, days, budget=100000.0, graph=None):
        """`days` days of the device-resident loop in one call; graph=True replays pairs of days from a captured
        hipGraph (same results, measured no faster: tools/experiments/measure_small_loop.py)"""
        if graph is not None:
            check(self._lib.adc_engine_day_graph_enable(self._h, 1 if graph else 0))
        check(self._lib.adc_engine_run_days(self._h, self.POLICIES[policy], int(days), float(budget)))

    def metrics_akncp_ncp(self, days):
        """(AKNCP [N], NCP [N]) of the running episode, reduced on the device (per-env median over the keywords in LDS)"""
        a, b = np.zeros(self.num_envs, np.float64), np.zeros(self.num_envs, np.float64)
        check(self._lib.adc_engine_metrics_akncp_ncp(self._h, float(days), a.ctypes.data, b.ctypes.data))
        return a, b

    def metrics_read_nk(self, ideal=True):
        """per (env, keyword) sums: profit in dollars, and (if ideal) the ideal sum and the ideal sum with <= 0 -> 1"""
        shape = (self.num_envs, self.num_keywords)
        pc = np.zeros(shape, np.int64)
        si = np.zeros(shape, np.float64) if ideal else None
        sp = np.zeros(shape, np.float64) if ideal else None
        check(self._lib.adc_engine_metrics_read_nk(self._h, pc.ctypes.data, None if si is None else si.ctypes.data,
                                                   None if sp is None else sp.ctypes.data))
        return pc / 100.0, si, sp


class ShardedStepEngine:
    """One vector of envs on ONE device, held by several engines (each with its own HIP stream) and stepped together.

    A host-in / host-out step is PCIe-bound (4 B per keyword up, 20 B down, against a 0.2 ms kernel at 4096 x 256): with the
    envs split over a few engines, enqueued asynchronously from page-locked buffers (adc_engine_step_async), one part's
    transfers overlap another part's kernels.  Results are those of a single engine: random streams are keyed by the
    GLOBAL env id (the multi-GPU sharding invariant).  The I/O arrays are single [N, K] page-locked arrays; every part reads
    and writes its rows of them, so callers see the same buffers a StepEngine gives them."""

    def __init__(self, num_envs, num_keywords, model=MODEL_IMPLICIT, *, shards=4, env_id_base=0, compact_counts=False, **kw):
        self.num_envs, self.num_keywords, self.model = int(num_envs), int(num_keywords), int(model)
        if np.ndim(shards):          # relative sizes: a small first part puts its results on the bus early
            edges = np.concatenate([[0.0], np.cumsum(np.asarray(shards, dtype=np.float64))])
            self.bounds = sorted(set(int(round(b)) for b in edges / edges[-1] * self.num_envs))
        else:
            shards = max(1, min(int(shards), self.num_envs))
            self.bounds = [int(b) for b in np.linspace(0, self.num_envs, shards + 1)]
        self.parts = [StepEngine(b1 - b0, num_keywords, model, env_id_base=env_id_base + b0, **kw)
                      for b0, b1 in zip(self.bounds[:-1], self.bounds[1:])]
        N, K = self.num_envs, self.num_keywords
        alloc = self.parts[0]._pinned_array
        self.compact_counts = bool(compact_counts)
        # per-keyword outputs: planes of one [., N, K] page-locked array (a part's rows of all planes = one 2-D transfer);
        # per-env outputs: every part has its own small block laid out like its device block (one transfer), gathered into
        # the [N] arrays after the wait
        per_kw = [n for n, _, k in _OUT_SPEC if k and not (self.compact_counts and n in _COUNT_NAMES)]
        planes = alloc((len(per_kw), N, K), np.int32)
        self.out = {n: planes[i].view(dict((a, b) for a, b, _ in _OUT_SPEC)[n]) for i, n in enumerate(per_kw)}
        self._counts_u16, self._overflow = None, None
        if self.compact_counts:
            self._counts_u16, self._overflow = alloc((N, 3, K), np.uint16), alloc((len(self.parts),), np.int32)
            for i, n in enumerate(_COUNT_NAMES):
                self.out[n] = self._counts_u16[:, i, :]
        per_env = [n for n, _, k in _OUT_SPEC if not k]
        self.out.update({n: np.zeros(N, dt) for n, dt, k in _OUT_SPEC if not k})
        self._small = []
        for p in self.parts:
            off, total = _out_offsets(p._lib, p._h)
            self._small.append(_views_at(alloc((total - off[5],), np.uint8), [o - off[5] for o in off], p.num_envs, K, per_env))
        self._bids_stage, self._budget_stage = alloc((N, K), np.float32), alloc((N,), np.float32)
        self._flat_act, self._flat_obs = None, None
        self._in_ptrs = [(self._bids_stage[b0:b1].ctypes.data, self._budget_stage[b0:b1].ctypes.data)
                         for b0, b1 in zip(self.bounds[:-1], self.bounds[1:])]
        self._outs = []
        for i, (b0, b1) in enumerate(zip(self.bounds[:-1], self.bounds[1:])):
            o = _out_struct(self.out, self._counts_u16, self._overflow, b0, b1, self._small[i])
            if self._overflow is not None:
                o.counts_overflow = self._overflow[i:i + 1].ctypes.data
            self._outs.append(o)

    def _each(self):
        return zip(self.parts, self.bounds[:-1], self.bounds[1:])

    def __getattr__(self, name):
        # (only reached for names this class does not define: the device-resident agent / metric / curve calls of StepEngine)
        if hasattr(StepEngine, name):
            raise NotImplementedError(f"{name}() works on one engine: construct the env / engine with engine_shards=1 "
                                      "(several engines per device only serve the host-in / host-out step)")
        raise AttributeError(name)

    def close(self):
        self.out, self._bids_stage, self._budget_stage, self._flat_act, self._flat_obs = {}, None, None, None, None
        self._counts_u16, self._overflow, self._small = None, None, []
        for p in self.parts:
            p.close()
        self.parts = []

    def set_all_params(self, planes):
        for p, b0, b1 in self._each():
            p.set_all_params(np.ascontiguousarray(planes[:, b0:b1]))

    def get_all_params(self):
        return np.concatenate([p.get_all_params() for p in self.parts], axis=1)

    def reset(self, env_mask=None, seeds=None):
        for p, b0, b1 in self._each():
            p.reset(None if env_mask is None else np.asarray(env_mask)[b0:b1], None if seeds is None else np.asarray(seeds)[b0:b1])

    def generate_keywords(self, table, no_vol_prob=0.0, env_mask=None, serial=0):
        for p, b0, b1 in self._each():
            p.generate_keywords(table, no_vol_prob, None if env_mask is None else np.asarray(env_mask)[b0:b1], serial)

    def generate_explicit_keywords(self, env_mask=None, serial=0):
        for p, b0, b1 in self._each():
            p.generate_explicit_keywords(None if env_mask is None else np.asarray(env_mask)[b0:b1], serial)

    def set_limits(self, max_days, loss_threshold):
        for p in self.parts:
            p.set_limits(max_days, loss_threshold)

    def get_rng_state(self):
        ks, ts = zip(*(p.get_rng_state() for p in self.parts))
        return np.concatenate(ks), np.concatenate(ts)

    def synchronize(self):
        for p in self.parts:
            p.synchronize()

    def step(self, bids, budget, copy=True):
        """host in / host out: every part's step is enqueued, then all are awaited"""
        stage_bids = not _is_buffer(bids, self._bids_stage)      # action_buffers() filled in place: nothing to stage
        if stage_bids and np.ndim(bids):
            bids = np.asarray(bids, dtype=np.float32).reshape(-1, self.num_keywords)
        if not _is_buffer(budget, self._budget_stage):
            self._budget_stage[...] = budget
        for (p, b0, b1), o, (pb, pg) in zip(self._each(), self._outs, self._in_ptrs):
            if stage_bids:                              # part by part: the next part is staged while this one's transfer runs
                self._bids_stage[b0:b1] = bids[b0:b1] if np.ndim(bids) else bids
            p.step_async(pb, pg, o)
        self._wait_and_gather()
        _check_overflow(self._overflow)
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out

    def _wait_and_gather(self):
        for p in self.parts:
            p.wait()
        for small, b0, b1 in zip(self._small, self.bounds[:-1], self.bounds[1:]):
            for name, a in small.items():
                self.out[name][b0:b1] = a

    def action_buffers(self):
        """(bids [N, K], budget [N]) page-locked arrays: fill them in place and pass them to step() to skip the staging copy"""
        return self._bids_stage, self._budget_stage

    def step_flat(self, flat_actions):
        N, K = self.num_envs, self.num_keywords
        if self._flat_act is None:
            alloc = self.parts[0]._pinned_array
            self._flat_act, self._flat_obs = alloc((N, K + 1), np.float32), alloc((N, 5 * K + 2), np.float32)
        o = self.out
        flat_actions = np.asarray(flat_actions, dtype=np.float32).reshape(N, K + 1)
        for (p, b0, b1), small in zip(self._each(), self._small):
            if not _is_buffer(flat_actions, self._flat_act):
                self._flat_act[b0:b1] = flat_actions[b0:b1]
            p.step_flat_async(self._flat_act[b0:b1].ctypes.data, self._flat_obs[b0:b1].ctypes.data, small["reward"].ctypes.data,
                              small["terminated"].ctypes.data, small["truncated"].ctypes.data)
        self._wait_and_gather()
        return self._flat_obs, o["reward"], o["terminated"], o["truncated"]

    def step_device(self, d_bids=None, d_budget=None):
        if d_bids is not None or d_budget is not None:
            raise NotImplementedError("caller-owned device actions need one engine (engine_shards=1)")
        for p in self.parts:
            p.step_device()


class ReplayTape:
    """Host-side tape for StepEngine.step_replay: the variates the reference drew, in its order."""

    def __init__(self, num_envs, volumes, bid_cents=(), click=(), conv=(), rev_cents=(), x_impressions=(), x_cost=(),
                 offsets=None, drift_uniforms=None):
        N = int(num_envs)
        self.vol = np.ascontiguousarray(volumes, dtype=np.int32)
        self.bid = np.ascontiguousarray(bid_cents, dtype=np.int32)
        self.click = np.ascontiguousarray(click, dtype=np.uint8)
        self.conv = np.ascontiguousarray(conv, dtype=np.uint8)
        self.rev = np.ascontiguousarray(rev_cents, dtype=np.int32)
        self.ximp = np.ascontiguousarray(x_impressions, dtype=np.int32)
        self.xcost = np.ascontiguousarray(x_cost, dtype=np.float64)
        names = ("bid", "ximp", "xcost", "click", "conv", "rev")
        offsets = offsets or {}
        self.off = {n: np.ascontiguousarray(offsets.get(n, np.zeros(N)), dtype=np.int64) for n in names}
        self.end = {n: np.zeros(N, dtype=np.int64) for n in names}
        self.struct = _ffi.Tape(
            self.vol.ctypes.data, self.bid.ctypes.data, self.ximp.ctypes.data, self.xcost.ctypes.data,
            self.click.ctypes.data, self.conv.ctypes.data, self.rev.ctypes.data,
            self.bid.size, self.ximp.size, self.xcost.size, self.click.size, self.conv.size, self.rev.size,
            *(self.off[n].ctypes.data for n in names), *(self.end[n].ctypes.data for n in names), None)
        # the three vectors update_keywords() drew after this step (vol, ctr, cvr; gymnasium_kw_env.py:132-135), [3][N][K]
        self.drift = None if drift_uniforms is None else np.ascontiguousarray(drift_uniforms, dtype=np.float32)
        if self.drift is not None:
            assert self.drift.size == 3 * self.vol.size, "drift_uniforms must be [3][num_envs][num_keywords]"
            self.struct.drift_uniforms = self.drift.ctypes.data
