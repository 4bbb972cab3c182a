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


class StepEngine:
    def __init__(self, num_envs, num_keywords, model=MODEL_IMPLICIT, *, device_id=0, max_days=60,
                 loss_threshold=10000.0, drift=(0.03, 0.03, 0.03), drift_enabled=False, impression_thresh=0.05,
                 auto_reset=False, env_id_base=0, seed=0):
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
        self.out = {name: self._pinned_array((N, K) if per_kw else (N,), dt) for name, dt, per_kw in _OUT_SPEC}
        self._out = _ffi.StepOut(*(self.out[name].ctypes.data for name, _, _ in _OUT_SPEC))
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
        self._bids_stage[...] = np.asarray(bids, dtype=np.float32).reshape(-1, self.num_keywords) if np.ndim(bids) else bids
        self._budget_stage[...] = budget
        return self._bids_stage, self._budget_stage

    def step(self, bids, budget, copy=True):
        """host in / host out, synchronous.  Returns dict of numpy arrays (views of reused buffers if copy=False)."""
        b, g = self._actions(bids, budget)
        check(self._lib.adc_engine_step(self._h, b.ctypes.data, g.ctypes.data, C.byref(self._out)))
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out

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

    def step_replay(self, bids, budget, tape, copy=True):
        b, g = self._actions(bids, budget)
        check(self._lib.adc_engine_step_replay(self._h, b.ctypes.data, g.ctypes.data, C.byref(tape.struct), C.byref(self._out)))
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out

    def step_device(self, d_bids=None, d_budget=None):
        """asynchronous; None = the engine's staging buffers (see sample_actions / device_buffer)."""
        check(self._lib.adc_engine_step_device(self._h, d_bids, d_budget))

    def fetch(self, copy=True):
        check(self._lib.adc_engine_fetch(self._h, C.byref(self._out)))
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
    def profile_enable(self, on=True):
        check(self._lib.adc_engine_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """(ms_fast_pass, ms_exact_pass_and_tail, ms_metric_accumulate), steps - summed over the steps since the last read"""
        ms = (C.c_double * 3)()
        n = C.c_int64()
        check(self._lib.adc_engine_profile_read(self._h, ms, C.byref(n)))
        return tuple(ms), n.value

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
        hipGraph (same results, measured no faster: tools/measure_small_loop.py)"""
        if graph is not None:
            check(self._lib.adc_engine_day_graph_enable(self._h, 1 if graph else 0))
        check(self._lib.adc_engine_run_days(self._h, self.POLICIES[policy], int(days), float(budget)))

    def metrics_read_nk(self, ideal=True):
        """per (env, keyword) sums: profit in dollars, and (if ideal) the ideal sum and the ideal sum with <= 0 -> 1"""
        shape = (self.num_envs, self.num_keywords)
        pc = np.zeros(shape, np.int64)
        si = np.zeros(shape, np.float64) if ideal else None
        sp = np.zeros(shape, np.float64) if ideal else None
        check(self._lib.adc_engine_metrics_read_nk(self._h, pc.ctypes.data, None if si is None else si.ctypes.data,
                                                   None if sp is None else sp.ctypes.data))
        return pc / 100.0, si, sp


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
