"""BiddingSimulation - the Gymnasium environment of adcraft/gymnasium_kw_env.py with its step path on the GPU.

Same constructor keywords, reset()/step()/render()/close()/set_updater_mask()/update_keywords(), same
attributes (action_space, observation_space, num_keywords, budget, max_days, loss_threshold,
keyword_params, keywords, current_day, cumulative_profit, updater_params, updater_mask, np_random,
metadata) and the same assertion behaviour (reference lines cited inline).  One instance is an
N=1 engine; for thousands of environments use vector_env.BiddingSimulationVectorEnv, which is ONE
engine call per step for all of them.

What differs from the reference, on purpose (DESIGN.md "quirks"):
  * step-time randomness is the engine's Philox stream keyed by the reset seed (the reference's own
    stream - numpy PCG64 interleaved with an UNSEEDED Rust thread_rng - is not reproducible even by
    the reference); keyword parameters generated at reset(seed) ARE bit-identical to the reference's;
  * bids are canonicalised to integer cents (numpy-1.x promotion semantics, SURVEY B-9);
  * info["bidding_outcomes"] is formatted only when somebody reads it: the per-click lists of src/lib.rs:251-275 never
    exist in a fused kernel, but every variate is addressed by what it is for, so they are regenerated exactly by a
    read-only second walk of that step (adc_engine_outcomes_replay).  step() itself does nothing for it.  An info dict
    read late (after a reset, a parameter change, or - with drift on - after the next step) carries the step's
    per-keyword totals instead and says 'per_click': 'expired';
  * only updater_mask None or all-True is supported (the only masks the reference's configs use; a
    partial mask mis-aligns coefficients in the reference, gymnasium_kw_env.py:136-144).
"""
import weakref
from typing import Dict, List, Optional

import numpy as np

from . import gymnasium_kw_utils as utils
from . import _ffi
from . import spaces as _spaces
from ._ffi import MODEL_EXPLICIT, MODEL_IMPLICIT, P_BCTR, P_SCTR, P_VOL_MEAN

try:  # pragma: no cover - depends on the image
    import gymnasium as _gym
    _EnvBase = _gym.Env
except Exception:
    class _EnvBase:
        """the slice of gymnasium.Env this environment relies on: np_random + reset(seed) seeding"""
        metadata: dict = {}
        _np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
            return self._np_random

        @np_random.setter
        def np_random(self, value):
            self._np_random = value

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

        @property
        def unwrapped(self):
            return self


class _Lazy:
    """a string that is only formatted when somebody looks at it"""

    def __init__(self, fn):
        self._fn, self._s = fn, None

    def __str__(self):
        if self._s is None:
            self._s = self._fn()
            self._fn = None
        return self._s

    __repr__ = __str__

    def __eq__(self, other):
        return str(self) == str(other)

    def __len__(self):
        return len(str(self))


class KeywordView:
    """Read-only view of one keyword's current parameters, with the few methods experiment code calls on
    the reference's Keyword objects (buyside_ctr, sellside_paid_ctr, sample_bids, sample_volume)."""

    def __init__(self, env, index):
        self._env, self._k = env, index

    def _p(self):
        return self._env.keyword_params[self._k]

    @property
    def rng(self):
        return self._env.np_random

    @property
    def buyside_ctr(self):
        return self._p()[3]

    @property
    def sellside_paid_ctr(self):
        return self._p()[4]

    def sample_bids(self, num_auctions=1):
        """ImplicitKeyword.sample_bids for the env's single-competitor keywords
        (adcraft/synthetic_kw_classes.py:610-621, gymnasium_kw_utils.py:183-184): shape (1, n)"""
        p = self._p()
        if not self._env._implicit:
            raise AttributeError("ExplicitKeyword has no sample_bids")
        loc, scale = p[1], 1.0 / p[2]
        return np.around(np.maximum(np.abs(self.rng.laplace(loc, scale, (1, num_auctions))), 0.0).astype(float), 2)

    def sample_volume(self, n=1):
        m, s = self._p()[0]
        return np.array([int(np.floor(max(self.rng.normal(m, s), 0.0) + 0.5)) for _ in range(n)])


class BiddingSimulation(_EnvBase):
    metadata = {"render_modes": ["ansi"]}

    def __init__(self, keyword_config: Optional[Dict] = None, num_keywords: int = 10, budget: float = 1000.0,
                 render_mode: Optional[str] = None, loss_threshold: float = 10000.0, max_days: int = 60,
                 updater_params: List[List] = [["vol", 0.03], ["ctr", 0.03], ["cvr", 0.03]],
                 updater_mask: Optional[List[bool]] = None, **kwargs) -> None:
        super().__init__()
        self.keyword_config = keyword_config
        self.num_keywords = num_keywords
        self.budget = budget
        self.action_space = _spaces.get_action_space(self.num_keywords)
        self.observation_space = _spaces.get_observation_space(self.num_keywords, self.budget)
        self.max_days = max_days
        self.loss_threshold = loss_threshold
        self.metadata = {"render_modes": ["ansi"]}
        assert render_mode is None or render_mode in self.metadata["render_modes"], (
            f'Specified render_mode of ({render_mode}) is not in the allowed options of '
            f'({", ".join(self.metadata["render_modes"])})')                       # gymnasium_kw_env.py:91-93
        self.render_mode = render_mode
        self._have_keywords = False
        self._current_text = "New start\n"
        self.updater_params = updater_params
        self.updater_mask = None
        self.init_volumes = None
        self._device_id = int(kwargs.get("device_id", 0))      # every other extra kwarg is swallowed, like the
        self._implicit = keyword_config is not None            # reference's **kwargs (multi_agent/env.py:31)
        self._engine = None
        self._params_host = None          # the reference-format keyword_params list (mutable lists)
        self._params_dirty = False
        self._serial = 0                  # steps taken since the stream was last (re)keyed or the parameters last changed
        self._epoch = 0                   # bumped by whatever ends the replayability of earlier steps
        self._last_outcomes = None        # weak reference to the last step's unread info["bidding_outcomes"]
        self.current_day = 0
        self.cumulative_profit = 0.0
        if updater_mask is not None:
            self.set_updater_mask(updater_mask)

    # ------------------------------------------------------------------ drift controls
    def set_updater_mask(self, new_updater_mask: List[bool]) -> None:
        assert len(new_updater_mask) == self.num_keywords, (
            f"Updater mask length ({len(new_updater_mask)})\n"
            + "must match number of keywords ({self.num_keywords}) to be applied.")   # gymnasium_kw_env.py:107-110
        m = [bool(x) for x in new_updater_mask]
        if any(m) and not all(m):
            raise NotImplementedError("only updater_mask=None or all-True is supported (see module docstring)")
        self.updater_mask = list(new_updater_mask)
        self.num_updates = int(np.sum(self.updater_mask))
        if self._engine is not None:
            self._engine.set_drift(all(m) and len(m) > 0, self._drift_coeffs())

    def _drift_coeffs(self):
        return tuple(float(v[1]) for v in self.updater_params)

    def _drift_on(self):
        return self.updater_mask is not None and len(self.updater_mask) > 0 and all(self.updater_mask)

    def update_keywords(self) -> None:
        """gymnasium_kw_env.py:114-158, on the device (engine drift stream)"""
        if self.updater_mask is None:
            return
        assert len(self.updater_mask) == self.num_keywords
        assert self._engine is not None, "reset required, need to generate keywords to bid on"
        self._settle_last_outcomes()
        self._engine.update_keywords()
        self._params_dirty = True
        self._epoch += 1

    def _settle_last_outcomes(self):
        """With drift on, the last step's update_keywords() is still pending on the device and its per-click lists are regenerated
        from the parameters that step ran with.  Whatever is about to write the drifted parameters into the planes (reading
        keyword_params, update_keywords()) first formats the last step's info["bidding_outcomes"] - if that info dict is still
        alive and unread - so the text a caller reads afterwards is that step's own."""
        ref = self._last_outcomes
        lazy = ref() if ref is not None else None
        if lazy is not None and self._drift_on():
            str(lazy)
        self._last_outcomes = None

    # ------------------------------------------------------------------ keyword state
    @property
    def keyword_params(self):
        if self._params_dirty and self._engine is not None:
            self._settle_last_outcomes()
            planes = self._engine.get_all_params()[:, 0, :].astype(np.float64)
            for k, p in enumerate(self._params_host):
                p[0] = (float(planes[P_VOL_MEAN, k]), p[0][1])
                p[3] = float(planes[P_BCTR, k])
                p[4] = float(planes[P_SCTR, k])
            self._params_dirty = False
        return self._params_host

    @keyword_params.setter
    def keyword_params(self, value):
        self._params_host = [list(p) for p in value]
        self._upload_params()

    def _upload_params(self):
        if self._implicit:
            raw = [(p[0], p[1], 1.0 / p[2], p[3], p[4], p[5], p[6]) for p in self._params_host]
            planes = utils.implicit_params_to_planes(raw)
        else:
            planes = utils.explicit_params_to_planes([tuple(p) for p in self._params_host])
        self._engine.set_env_params(0, planes)
        self._params_dirty = False
        self._epoch += 1

    def _ensure_engine(self, seed):
        if self._engine is None:
            from .engine import StepEngine      # raises if the HIP library / device is missing: no fallback
            self._engine = StepEngine(1, self.num_keywords, MODEL_IMPLICIT if self._implicit else MODEL_EXPLICIT,
                                      device_id=self._device_id, max_days=self.max_days,
                                      loss_threshold=self.loss_threshold, drift=self._drift_coeffs(),
                                      drift_enabled=self._drift_on(), seed=0 if seed is None else seed)

    # ------------------------------------------------------------------ reset / step
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self._epoch += 1
        resample = seed is not None or not self._have_keywords          # gymnasium_kw_env.py:303
        self._ensure_engine(seed)
        if resample:
            if self.keyword_config is not None:
                raw = utils.sample_implicit_keyword_params(self.num_keywords, self.np_random, self.keyword_config)
            else:
                raw = utils.sample_random_keyword_params(self.num_keywords, self.np_random)
            utils.consume_construction_draws(self.num_keywords, self.np_random)
            self._params_host = [list(p) for p in utils.printed_params(raw, self._implicit)]   # :315
            self.keywords = [KeywordView(self, k) for k in range(self.num_keywords)]
            self._upload_params()
            self._have_keywords = True
            stream_seed = seed if seed is not None else int(self.np_random.integers(0, 2**63 - 1))
            self._engine.reset(seeds=np.array([stream_seed], dtype=np.uint64))
        else:
            self._engine.reset()                                        # keeps keywords and any drift (B-7)
        if options:                                                     # :318-325
            self.max_days = options.get("max_days", self.max_days)
            rm = options.get("render_mode", self.render_mode)
            if rm is None or rm in self.metadata["render_modes"]:
                self.render_mode = rm
            self.loss_threshold = options.get("loss_threshold", self.loss_threshold)
            self._engine.set_limits(self.max_days, self.loss_threshold)
        self.current_day = 0
        self.cumulative_profit = 0.0
        self._current_text = "Reset environment\n\nNew start\n"
        K = self.num_keywords
        observations = dict(impressions=np.zeros(K, dtype=int), buyside_clicks=np.zeros(K, dtype=int),
                            cost=np.zeros(K, dtype=np.float32), sellside_conversions=np.zeros(K, dtype=int),
                            revenue=np.zeros(K, dtype=np.float32), cumulative_profit=np.zeros(1, dtype=np.float32),
                            days_passed=np.zeros(1, dtype=np.float32))     # typed zeros (:340-342, B-13)
        info = {"keyword_params": utils.repr_all_params(self.keyword_params)}
        return observations, info

    def step(self, action):
        assert self._have_keywords, "reset required, need to generate keywords to bid on"   # :194-196
        budget_array = action.get("budget", self.budget)
        bid_array = action.get("keyword_bids")
        self.budget = np.round(budget_array, 2).astype(float)                                # :199
        bids64 = np.asarray(bid_array, dtype=np.float64).reshape(self.num_keywords)
        rounded = np.round(np.maximum(bids64, 0.01), 2)                                      # :215 (f64: B-9)
        bids = rounded.tolist()
        out = self._engine.step(rounded.astype(np.float32)[None, :],
                                np.float32(np.asarray(self.budget, dtype=np.float64).reshape(-1)[0]), copy=False)
        profits = float(out["reward"][0])
        self.cumulative_profit = float(out["cumulative_profit"][0])
        truncated = bool(out["truncated"][0])
        self.current_day = int(out["days_passed"][0])
        terminated = bool(out["terminated"][0])
        reward = profits
        observations = dict(
            impressions=out["impressions"][0].astype(int),
            buyside_clicks=out["buyside_clicks"][0].astype(int),
            cost=out["cost"][0].copy(),
            sellside_conversions=out["sellside_conversions"][0].astype(int),
            revenue=out["revenue"][0].copy(),
            cumulative_profit=np.array([self.cumulative_profit], dtype=np.float32),
            days_passed=np.array([self.current_day], dtype=np.float32))
        if self._drift_on():
            self._params_dirty = True            # update_keywords() ran on the device (:246)
        snap = dict(observations)            # (the arrays are this step's own copies; formatted only if somebody looks)
        budget_used = float(np.float32(np.asarray(self.budget, dtype=np.float64).reshape(-1)[0]))
        self._serial += 1
        serial, epoch = self._serial, self._epoch
        outcomes = _Lazy(lambda: self._outcomes_text(serial, epoch, bids, rounded.astype(np.float32), budget_used, snap))
        self._last_outcomes = weakref.ref(outcomes)
        info = {
            "bids": bids,
            "bidding_outcomes": outcomes,
            "keyword_params": _Lazy(lambda: utils.repr_all_params(self.keyword_params)),
        }
        if self.render_mode == "ansi":                                                       # :253-260
            self._current_text = (
                f"Time step: {self.current_day}/{self.max_days},   "
                + f"Average profit per kw in step: {profits/self.num_keywords:.2f},   "
                + f"Budget: {self.budget}   "
                + f"Total profit in step: {profits:.2f},   "
                + f"Cumulative profit: {self.cumulative_profit:.2f}\n")
        if truncated:                                                                        # :262-267
            self._current_text += ("Bidding simulation truncated early, we spent too much.\n"
                                   + f"Our allowed spend was ({self.loss_threshold:.2f}),\n"
                                   + f"but our cumulative loss was ({self.cumulative_profit:.2f})")
        return observations, reward, terminated, truncated, info

    def _outcomes_text(self, serial, epoch, bids, bids_f32, budget_used, obs):
        """info["bidding_outcomes"] of the step with this serial, formatted when somebody reads it.  The per-click lists are
        regenerated by replaying that step's stream position read-only (StepEngine.outcomes_replay); nothing is done at step
        time, so a loop that never looks pays nothing.  The last step can always be replayed; an earlier one while its episode
        is still running, drift is off and the keyword parameters have not been touched since.  Past that the clicks cannot be
        regenerated: the text then carries the step's per-keyword totals (exact) with one-element lists, and says so under
        'per_click'."""
        back = self._serial - serial + 1
        replayable = (self._engine is not None and epoch == self._epoch and (back == 1 or not self._drift_on()))
        if replayable:
            try:
                clicks = self._engine.outcomes_replay(0, bids_f32, budget_used, steps_back=back)
            except _ffi.EngineStateError:   # ADC_ESTATE only: the engine itself says this step's clicks are gone (e.g. its drift was applied)
                clicks = None
            return self._repr_outcomes(bids, obs, clicks)
        return self._repr_outcomes(bids, obs, None)

    @staticmethod
    def _repr_outcomes(bids, obs, clicks):
        """src/lib.rs:251-275 (repr_outcomes_py) on the regenerated per-click lists: `{}` of an f64 prints 1 for 1.0, `{:?}`
        prints 1.0; 'profit' adds the sub-timesteps' (revenues - costs) in order, as combine_outcomes does
        (bidding_simulation.py:117-119,127-133)"""
        def disp(x):
            x = float(x)
            return str(int(x)) if x == int(x) and abs(x) < 1e16 else repr(x)

        rows = []
        if clicks is None:                                  # expired (see _outcomes_text): the step's totals only
            for k, b in enumerate(bids):
                c, r = float(obs["cost"][k]), float(obs["revenue"][k])
                rows.append("{" + f"'bid': {disp(b)}, 'impressions': {int(obs['impressions'][k])}, "
                            f"'buyside_clicks': {int(obs['buyside_clicks'][k])}, 'costs': [{c!r}], "
                            f"'sellside_conversions': {int(obs['sellside_conversions'][k])}, 'revenues': [{r!r}], "
                            f"'profit': {disp(r - c)}, 'per_click': 'expired'" + "}")
            return "[" + ", ".join(rows) + "]"
        for o in combined_outcomes(bids, obs, clicks):
            rows.append("{" + f"'bid': {disp(o['bid'])}, 'impressions': {o['impressions']}, 'impression_share': {disp(o['impression_share'])}, "
                        f"'buyside_clicks': {o['buyside_clicks']}, 'costs': {o['costs']!r}, "
                        f"'sellside_conversions': {o['sellside_conversions']}, 'revenues': {o['revenues']!r}, "
                        f"'revenues_per_cost': {o['revenues_per_cost']!r}, 'profit': {disp(o['profit'])}" + "}")
        return "[" + ", ".join(rows) + "]"

    def render(self) -> Optional[str]:
        if self.render_mode == "ansi":
            return self._current_text

    def close(self):
        self._epoch += 1
        if self._engine is not None:
            self._engine.close()
            self._engine = None


def combined_outcomes(bids, obs, clicks):
    """The step's combined BiddingOutcomes (bidding_simulation.py:10-38), one dict per keyword, from the regenerated click records
    (StepEngine.outcomes_replay: keyword, timestep, cost, revenue | -1 per paid click in the reference's order, share_volume):
    'costs' / 'revenues' / 'revenues_per_cost' are the click lists as combine_outcomes concatenates them (:138-141), 'profit' adds
    each sub-timestep's sum(revenues) - sum(costs) in order (:117,136-137), 'impression_share' = impressions over the auctions of
    the sub-timesteps that had an impression - the denominator combine_outcomes' np.round(impressions / impression_share) chain
    ends up with (:128-145; a sub-timestep without impressions drops out of it)."""
    K = len(bids)
    by_kw = [[] for _ in range(K)]
    for i in range(len(clicks["keyword"])):
        by_kw[int(clicks["keyword"][i])].append(i)
    out = []
    for k, b in enumerate(bids):
        idx = by_kw[k]
        costs = [float(clicks["cost"][i]) for i in idx]
        rpc = [max(float(clicks["revenue"][i]), 0.0) for i in idx]
        revenues = [float(clicks["revenue"][i]) for i in idx if clicks["revenue"][i] >= 0]
        profit, t_prev, cell_r, cell_c = 0.0, None, 0.0, 0.0
        for i in idx:                                   # per sub-timestep: sum(revenues) - sum(costs), then added up
            t = int(clicks["timestep"][i])
            if t != t_prev and t_prev is not None:
                profit += cell_r - cell_c
                cell_r = cell_c = 0.0
            t_prev = t
            cell_c += float(clicks["cost"][i])
            if clicks["revenue"][i] >= 0:
                cell_r += float(clicks["revenue"][i])
        if t_prev is not None:
            profit += cell_r - cell_c
        imp = int(obs["impressions"][k])
        vol = int(clicks["share_volume"][k])
        out.append(dict(bid=float(b), impressions=imp, impression_share=imp / vol if vol > 0 else 0.0,
                        buyside_clicks=int(obs["buyside_clicks"][k]), costs=costs,
                        sellside_conversions=int(obs["sellside_conversions"][k]), revenues=revenues, revenues_per_cost=rpc, profit=profit))
    return out


def bidding_sim_creator(env_config: Dict) -> BiddingSimulation:
    """gymnasium_kw_env.py:361-363"""
    return BiddingSimulation(**env_config)
