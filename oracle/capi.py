"""ctypes binding of oracle/adcraft_oracle.c (TEST INFRASTRUCTURE - never imported by the product)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
IMPLICIT, EXPLICIT, IMPLICIT_GENERAL = 0, 1, 2
P_VOL_MEAN, P_VOL_STD, P_A, P_B, P_BCTR, P_SCTR, P_REV_MEAN, P_REV_STD, P_COUNT = range(9)


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


class Config(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_keywords", C.c_int32), ("model", C.c_int32), ("max_days", C.c_int32),
                ("loss_threshold", C.c_double),
                ("drift_vol", C.c_float), ("drift_ctr", C.c_float), ("drift_cvr", C.c_float),
                ("drift_on", C.c_int32), ("imp_thresh", C.c_float), ("auto_reset", C.c_int32), ("threads", C.c_int32),
                ("max_bidders", C.c_int32), ("participation_rate", C.c_float), ("num_winners", C.c_int32)]


class Tape(C.Structure):
    _fields_ = [("volumes", C.c_void_p), ("bid_cents", C.c_void_p), ("x_impressions", C.c_void_p),
                ("x_cost", C.c_void_p), ("click", C.c_void_p), ("conv", C.c_void_p), ("rev_cents", C.c_void_p),
                ("cur_bid", C.c_int64), ("cur_ximp", C.c_int64), ("cur_xcost", C.c_int64),
                ("cur_click", C.c_int64), ("cur_conv", C.c_int64), ("cur_rev", C.c_int64),
                ("drift_uniforms", C.c_void_p)]


class Out(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("impressions", "clicks", "conversions", "cost_cents", "revenue_cents",
                                          "cost", "revenue", "volumes", "reward", "cum_profit", "day",
                                          "terminated", "truncated")]


class Outcomes(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("count", C.c_int64)] + [(n, C.c_void_p) for n in (
        "env", "keyword", "timestep", "cost", "revenue", "impression_share", "profit")]


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("params", "key", "tick", "day", "cum_cents", "cum", "drift_pending")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        plain, fma = _build.build()
        _lib = C.CDLL(fma if _cpu_has_fma() else plain)
        L = _lib
        L.orc_philox4x32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_philox4x32_r.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_det_logf.restype = C.c_float
        L.orc_det_logf.argtypes = [C.c_float]
        L.orc_det_expf.restype = C.c_float
        L.orc_det_expf.argtypes = [C.c_float]
        L.orc_normal_from_word.restype = C.c_float
        L.orc_normal_from_word.argtypes = [C.c_uint32]
        L.orc_laplace_cents_from_word.restype = C.c_int32
        L.orc_laplace_cents_from_word.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.orc_normal_tab.restype = C.c_float
        L.orc_normal_tab.argtypes = [C.c_uint32]
        L.orc_revenue_cents_tab.restype = C.c_int32
        L.orc_revenue_cents_tab.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.orc_check_deviate_monotone.restype = C.c_int64
        L.orc_check_deviate_monotone.argtypes = []
        L.orc_competitor_cents_from_v.restype = C.c_int32
        L.orc_competitor_cents_from_v.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.orc_neg_log_u24.restype = C.c_float
        L.orc_neg_log_u24.argtypes = [C.c_uint32]
        L.orc_auction_outcome.restype = C.c_int32
        L.orc_auction_outcome.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]
        L.orc_bernoulli_threshold.restype = C.c_uint64
        L.orc_bernoulli_threshold.argtypes = [C.c_float]
        L.orc_revenue_cents_from_word.restype = C.c_int32
        L.orc_revenue_cents_from_word.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.orc_volume_from_word.restype = C.c_int32
        L.orc_volume_from_word.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.orc_bid_cents.restype = C.c_int64
        L.orc_bid_cents.argtypes = [C.c_float]
        L.orc_check_div100f.restype = C.c_int64
        L.orc_check_div100f.argtypes = [C.c_int64, C.c_int64]
        L.orc_check_div100.restype = C.c_int64
        L.orc_check_div100.argtypes = [C.c_int64, C.c_int64]
        L.orc_budget_cents.restype = C.c_int64
        L.orc_budget_cents.argtypes = [C.c_float]
        L.orc_threshold_sigmoid_f32.restype = C.c_float
        L.orc_threshold_sigmoid_f32.argtypes = [C.c_float] * 4
        L.orc_threshold_sigmoid_f64.restype = C.c_double
        L.orc_threshold_sigmoid_f64.argtypes = [C.c_double] * 4
        L.orc_sigmoid_f64.restype = C.c_double
        L.orc_sigmoid_f64.argtypes = [C.c_double] * 3
        L.orc_explicit_cost_from_word.restype = C.c_float
        L.orc_explicit_cost_from_word.argtypes = [C.c_uint32, C.c_float]
        L.orc_nth_price_auction.restype = C.c_int32
        L.orc_nth_price_auction.argtypes = [C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_void_p]
        L.orc_step.restype = C.c_int32
        L.orc_step.argtypes = [C.POINTER(Config), C.POINTER(State), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Out)]
        L.orc_step_outcomes.restype = C.c_int32
        L.orc_step_outcomes.argtypes = [C.POINTER(Config), C.POINTER(State), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Out),
                                        C.POINTER(Outcomes)]
        L.orc_materialize_drift.restype = C.c_int32
        L.orc_materialize_drift.argtypes = [C.POINTER(Config), C.POINTER(State)]
        L.orc_sample_bids.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    return _lib


PHILOX_ROUNDS = 7        # the stream's round count (oracle/adcraft_oracle.c ORC_PHILOX_ROUNDS)


def philox(ctr, key, rounds=None):
    """Philox4x32 with the stream's round count (or `rounds`)"""
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    if rounds is None:
        lib().orc_philox4x32(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    else:
        lib().orc_philox4x32_r(c.ctypes.data, k.ctypes.data, int(rounds), o.ctypes.data)
    return o


def nth_price_auction(bid, other_bids, n=2, num_winners=2):
    ob = np.ascontiguousarray(other_bids, dtype=np.float64)
    na, nb = ob.shape
    pl = np.zeros(max(na, 1), dtype=np.int32)
    co = np.zeros(max(na, 1), dtype=np.float64)
    imp = lib().orc_nth_price_auction(float(bid), ob.ctypes.data, na, nb, n, num_winners, pl.ctypes.data, co.ctypes.data)
    return imp, pl[:imp].copy(), co[:imp].copy()


class OracleEngine:
    """Holds the caller-owned state arrays of orc_step for N envs x K keywords."""

    def __init__(self, num_envs, num_keywords, model=IMPLICIT, max_days=60, loss_threshold=10000.0,
                 drift=(0.03, 0.03, 0.03), drift_on=False, imp_thresh=0.05, auto_reset=False, threads=1,
                 max_bidders=30, participation_rate=0.6, num_winners=1):
        self.N, self.K = int(num_envs), int(num_keywords)
        self.cfg = Config(self.N, self.K, model, max_days, loss_threshold, drift[0], drift[1], drift[2],
                          1 if drift_on else 0, imp_thresh, 1 if auto_reset else 0, threads,
                          int(max_bidders), float(participation_rate), int(num_winners))
        N, K = self.N, self.K
        self.params = np.zeros((P_COUNT, N, K), dtype=np.float32)
        self.key = np.zeros(N, dtype=np.uint64)
        self.tick = np.zeros(N, dtype=np.uint32)
        self.day = np.zeros(N, dtype=np.int32)
        self.cum_cents = np.zeros(N, dtype=np.int64)
        self.cum = np.zeros(N, dtype=np.float64)
        self.drift_pending = np.zeros(N, dtype=np.uint8)
        self.state = State(*(a.ctypes.data for a in (self.params, self.key, self.tick, self.day, self.cum_cents,
                                                     self.cum, self.drift_pending)))
        self.out = dict(
            impressions=np.zeros((N, K), np.int32), clicks=np.zeros((N, K), np.int32),
            conversions=np.zeros((N, K), np.int32), cost_cents=np.zeros((N, K), np.int64),
            revenue_cents=np.zeros((N, K), np.int64), cost=np.zeros((N, K), np.float64),
            revenue=np.zeros((N, K), np.float64), volumes=np.zeros((N, K), np.int32),
            reward=np.zeros(N, np.float64), cum_profit=np.zeros(N, np.float64), day=np.zeros(N, np.int32),
            terminated=np.zeros(N, np.uint8), truncated=np.zeros(N, np.uint8))
        self._out = Out(*(self.out[n].ctypes.data for n, _ in Out._fields_))

    def step(self, bids, budget, tape=None):
        bids = np.ascontiguousarray(bids, dtype=np.float32).reshape(self.N, self.K)
        budget = np.ascontiguousarray(np.broadcast_to(np.asarray(budget, dtype=np.float32), (self.N,)))
        tp = None
        if tape is not None:
            tp = C.byref(tape.struct)
        rc = lib().orc_step(C.byref(self.cfg), C.byref(self.state), bids.ctypes.data, budget.ctypes.data, tp,
                            C.byref(self._out))
        assert rc == 0
        return {k: v.copy() for k, v in self.out.items()}

    def step_outcomes(self, bids, budget, tape=None, capacity=1 << 16):
        """step() plus the combined BiddingOutcomes the reference would hand to repr_outcomes_py: (out, outcomes) with outcomes =
        dict(costs, revenues, revenues_per_cost: [N][K] lists in the reference's order; impression_share, profit: [N, K])"""
        N, K = self.N, self.K
        bids = np.ascontiguousarray(bids, dtype=np.float32).reshape(N, K)
        budget = np.ascontiguousarray(np.broadcast_to(np.asarray(budget, dtype=np.float32), (N,)))
        while True:
            snapshot = [a.copy() for a in (self.params, self.key, self.tick, self.day, self.cum_cents, self.cum, self.drift_pending)]
            cursors = tape.cursors() if tape is not None else None
            env, kw, ts = (np.zeros(capacity, np.int32) for _ in range(3))
            cost, rev = np.zeros(capacity, np.float64), np.zeros(capacity, np.float64)
            share, profit = np.zeros((N, K), np.float64), np.zeros((N, K), np.float64)
            lists = Outcomes(capacity, 0, env.ctypes.data, kw.ctypes.data, ts.ctypes.data, cost.ctypes.data, rev.ctypes.data,
                             share.ctypes.data, profit.ctypes.data)
            rc = lib().orc_step_outcomes(C.byref(self.cfg), C.byref(self.state), bids.ctypes.data, budget.ctypes.data,
                                         C.byref(tape.struct) if tape is not None else None, C.byref(self._out), C.byref(lists))
            assert rc == 0
            if lists.count <= capacity:
                break
            capacity = int(lists.count)             # did not fit: put the state back and walk the step again
            for a, b in zip((self.params, self.key, self.tick, self.day, self.cum_cents, self.cum, self.drift_pending), snapshot):
                a[...] = b
            if tape is not None:
                for n, v in cursors.items():
                    setattr(tape.struct, "cur_" + n, v)
        m = int(lists.count)
        costs = [[[] for _ in range(K)] for _ in range(N)]
        revenues = [[[] for _ in range(K)] for _ in range(N)]
        rpc = [[[] for _ in range(K)] for _ in range(N)]
        for i in range(m):
            e, k = int(env[i]), int(kw[i])
            costs[e][k].append(float(cost[i]))
            rpc[e][k].append(max(float(rev[i]), 0.0))
            if rev[i] >= 0:
                revenues[e][k].append(float(rev[i]))
        out = {k: v.copy() for k, v in self.out.items()}
        return out, dict(costs=costs, revenues=revenues, revenues_per_cost=rpc, impression_share=share, profit=profit,
                         keyword=kw[:m].copy(), timestep=ts[:m].copy(), env=env[:m].copy(), cost=cost[:m].copy(), revenue=rev[:m].copy())

    def materialize_drift(self):
        lib().orc_materialize_drift(C.byref(self.cfg), C.byref(self.state))

    def sample_bids(self, lo=0.30, hi=1.00):
        b = np.zeros((self.N, self.K), dtype=np.float32)
        lib().orc_sample_bids(C.byref(self.cfg), self.key.ctypes.data, self.tick.ctypes.data, lo, hi, b.ctypes.data)
        return b


class TapeSource:
    """Flat variate tapes in the order the reference draws them; cursors persist across steps."""

    def __init__(self, bid_cents=(), click=(), conv=(), rev_cents=(), x_impressions=(), x_cost=()):
        self.bid = np.ascontiguousarray(bid_cents, dtype=np.int32)
        self.click = np.ascontiguousarray(click, dtype=np.uint8)
        self.conv = np.ascontiguousarray(conv, dtype=np.uint8)
        self.rev = np.ascontiguousarray(rev_cents, dtype=np.int32)
        self.ximp = np.ascontiguousarray(x_impressions, dtype=np.int32)
        self.xcost = np.ascontiguousarray(x_cost, dtype=np.float64)
        self.vol = None
        self.drift = None
        self.struct = Tape(0, self.bid.ctypes.data, self.ximp.ctypes.data, self.xcost.ctypes.data,
                           self.click.ctypes.data, self.conv.ctypes.data, self.rev.ctypes.data, 0, 0, 0, 0, 0, 0, None)

    def set_drift_uniforms(self, uniforms_3nk):
        """the three vectors update_keywords() drew (vol, ctr, cvr), [3][N][K]; None = no drift after the step"""
        self.drift = None if uniforms_3nk is None else np.ascontiguousarray(uniforms_3nk, dtype=np.float32)
        self.struct.drift_uniforms = None if self.drift is None else self.drift.ctypes.data

    def set_volumes(self, volumes):
        self.vol = np.ascontiguousarray(volumes, dtype=np.int32)
        self.struct.volumes = self.vol.ctypes.data

    def cursors(self):
        s = self.struct
        return dict(bid=s.cur_bid, ximp=s.cur_ximp, xcost=s.cur_xcost, click=s.cur_click, conv=s.cur_conv, rev=s.cur_rev)
