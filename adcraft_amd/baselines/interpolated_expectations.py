"""Mirror of adcraft/baselines/interpolated_expectations.py for the part the paper's experiments use:
``NaiveZeroMarginStrategy`` (:442-515).  The caches and the bid rule live in the HIP engine
(``k_agent_step``, adcraft_amd/csrc/parts/kernels_policy.inc), one agent per env; this class is the reference's
Python surface over it (same constructor, ``update_all_caches(prev_action, prev_observation)``, ``sample_action()``,
``caches``, ``max_bids``), usable with any env that returns the reference's observation dict.

For the closed loop without a host round trip per step use ``adcraft_amd.closed_loop`` instead.
"""
import numpy as np

from ..engine import StepEngine


def get_empty_cache():
    """adcraft/baselines/interpolated_expectations.py:286-295"""
    return {"ave_rpc": 0.0, "num_rpc_obs": 0, "ave_sctr": 0.4, "num_sctr_obs": 0.0, "ave_cpc": {}, "ave_clicks": {}}


class NaiveZeroMarginStrategy:
    """Estimates revenue per buyside click (rpc x sctr) and bids it; ramps the bid up while nothing has been observed
    (reference docstring :443-456).  ``engine`` (optional) is a StepEngine whose envs the agent serves: its device
    action buffers then receive the sampled action (``engine.step_device()`` consumes it)."""

    def __init__(self, num_keywords, default_expected_revenue_per_conversion=3.0, initial_caches=None, seed=None, *,
                 engine=None, device_id=0):
        if initial_caches is not None:
            raise NotImplementedError("preseeded caches are not supported by the device agent")
        self._own = engine is None
        self._e = engine if engine is not None else StepEngine(1, int(num_keywords), device_id=device_id)
        if self._e.num_keywords != int(num_keywords):
            raise ValueError("engine.num_keywords != num_keywords")
        self.observation_keys = ["impressions", "buyside_clicks", "cost", "sellside_conversions", "revenue"]
        self.default_rpc = default_expected_revenue_per_conversion
        # the reference draws from np.random.default_rng(seed); the device agent from its Philox stream keyed by `seed`
        self._e.agent_init(default_expected_revenue_per_conversion,
                           None if seed is None else np.full(self._e.num_envs, seed, dtype=np.uint64) + np.arange(self._e.num_envs, dtype=np.uint64))
        self.prev_bids = None

    def update_all_caches(self, prev_action, prev_observation):
        """:485-494; observations of shape [K] (one env) or [N, K]"""
        self.prev_bids = prev_action["keyword_bids"]
        self._e.agent_update(prev_observation["buyside_clicks"], prev_observation["sellside_conversions"], prev_observation["revenue"])

    def sample_action(self, replay_uniforms=None):
        """:496-515 -> {"budget", "keyword_bids"} (arrays squeezed for a single env)"""
        self._e.agent_act(0.0, replay_uniforms)
        bids, budget = self._e.get_actions()
        if self._e.num_envs == 1:
            return {"budget": float(budget[0]), "keyword_bids": bids[0].astype(np.float64)}
        return {"budget": budget.astype(np.float64), "keyword_bids": bids.astype(np.float64)}

    @property
    def max_bids(self):
        mb = self._e.agent_state()["max_bids"]
        return mb[0] if self._e.num_envs == 1 else mb

    @property
    def caches(self):
        """list (one env) or list of lists of the reference's cache dicts (ave_cpc / ave_clicks are not tracked: the
        zero-margin strategy never reads them)"""
        st = self._e.agent_state()

        def one(n):
            return [dict(ave_rpc=float(st["ave_rpc"][n, k]), num_rpc_obs=int(st["num_rpc_obs"][n, k]),
                         ave_sctr=float(st["ave_sctr"][n, k]), num_sctr_obs=float(st["num_sctr_obs"][n, k]),
                         ave_cpc={}, ave_clicks={}) for k in range(self._e.num_keywords)]
        return one(0) if self._e.num_envs == 1 else [one(n) for n in range(self._e.num_envs)]

    def close(self):
        if self._own:
            self._e.close()
