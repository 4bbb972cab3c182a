"""BiddingSimulationVectorEnv - N BiddingSimulation environments as ONE engine on one MI355X.

The reference's only vectorisation is RLlib's `num_envs_per_worker` (a serial Python loop over env copies,
adcraft/experiment_utils/agent_configs.py:60,85,107).  Here reset()/step() are batched: actions
[N, K] in, observations [N, K] out, one C-ABI call per step.  The class follows the
gymnasium.vector.VectorEnv calling convention (reset(seed=) -> (obs, infos), step(actions) ->
(obs, rewards, terminations, truncations, infos), single_action_space / single_observation_space,
same-step autoreset with the terminal episode totals in infos); thin adaptors below expose the RLlib
VectorEnv (vector_reset / reset_at / vector_step) and Stable-Baselines3 VecEnv (step_async /
step_wait) call shapes over the same engine.  Flat observations in FlatArrayWrapper order
(adcraft/wrappers/flat_array.py:74-80, adcraft/gymnasium_kw_utils.py:383-390) are available with
flat=True.
"""
from typing import Dict, Optional

import numpy as np

from . import gymnasium_kw_utils as utils
from . import spaces as _spaces
from . import synthetic
from ._ffi import MODEL_EXPLICIT, MODEL_IMPLICIT


class BiddingSimulationVectorEnv:
    metadata = {"render_modes": ["ansi"], "autoreset_mode": "same_step"}

    def __init__(self, num_envs: int, keyword_config: Optional[Dict] = None, num_keywords: int = 10,
                 budget: float = 1000.0, loss_threshold: float = 10000.0, max_days: int = 60,
                 updater_params=(("vol", 0.03), ("ctr", 0.03), ("cvr", 0.03)), updater_mask=None,
                 device_id: int = 0, env_id_base: int = 0, autoreset: bool = True, flat: bool = False,
                 param_sampler: str = "reference", copy: bool = False, engine_shards: Optional[int] = None, compact_counts: bool = False,
                 **kwargs):
        """param_sampler: "reference" draws every env's keywords with the reference's exact seeded recipe
        (env i uses seed + i; host loop, fine up to a few thousand envs); "device" draws the same law on the GPU
        from each env's own Philox key (no host loop, no upload; use it for 10^4+ envs); "vectorised" draws the
        same law for all envs at once with numpy.  All three serve both keyword models (keyword_config given: IMPLICIT keywords from
        the quantile tables; not given: the default constructor's EXPLICIT set, sample_random_keywords).
        engine_shards: the envs are held by this many engines on the device, stepped together so that one part's PCIe
        transfers overlap another's kernels (None: 4 from 2^18 keywords up, else 1); results do not depend on it.
        compact_counts: the three count observations cross PCIe as uint16 (packed on the device: 14 B instead of 20 B per
        keyword come back per step; step() raises OverflowError if a count exceeds 65535).  Dict observations only.
        copy: False (default) returns observation arrays that are views of the engine's page-locked I/O buffers -
        valid until the next step(), counts as int32 (the reference's int64 costs a 3x8 MB conversion per step at
        4096 x 256); True returns fresh arrays with the reference's dtypes (int64 counts)."""
        self.num_envs, self.num_keywords = int(num_envs), int(num_keywords)
        self.keyword_config = keyword_config
        self.budget = np.full(self.num_envs, float(budget), dtype=np.float32)
        self.loss_threshold, self.max_days = float(loss_threshold), int(max_days)
        self.updater_params = [list(p) for p in updater_params]
        if updater_mask is not None:
            assert len(updater_mask) == self.num_keywords
            if any(updater_mask) and not all(updater_mask):
                raise NotImplementedError("only updater_mask=None or all-True is supported")
        self.updater_mask = updater_mask
        self.single_action_space = _spaces.get_action_space(self.num_keywords)
        self.single_observation_space = _spaces.get_observation_space(self.num_keywords, float(budget))
        self.action_space, self.observation_space = self.single_action_space, self.single_observation_space
        self.flat = bool(flat)
        self.autoreset = bool(autoreset)
        self.param_sampler = param_sampler
        self.copy = bool(copy)
        self._implicit = keyword_config is not None
        self._device_id, self._env_id_base = int(device_id), int(env_id_base)
        self._shards = (4 if self.num_envs * self.num_keywords >= (1 << 18) else 1) if engine_shards is None else max(1, int(engine_shards))
        self._compact = bool(compact_counts) and not self.flat
        self._engine = None
        self._have_keywords = False

    # ------------------------------------------------------------------ engine
    def _ensure_engine(self, seed):
        if self._engine is None:
            from .engine import ShardedStepEngine, StepEngine
            drift_on = self.updater_mask is not None and len(self.updater_mask) > 0 and all(self.updater_mask)
            make = StepEngine if self._shards <= 1 else (lambda *a, **k: ShardedStepEngine(*a, shards=self._shards, **k))
            self._engine = make(self.num_envs, self.num_keywords,
                                      MODEL_IMPLICIT if self._implicit else MODEL_EXPLICIT,
                                      device_id=self._device_id, max_days=self.max_days,
                                      loss_threshold=self.loss_threshold,
                                      drift=tuple(float(p[1]) for p in self.updater_params), drift_enabled=drift_on,
                                      auto_reset=self.autoreset, env_id_base=self._env_id_base,
                                      seed=0 if seed is None else seed, compact_counts=self._compact)
        return self._engine

    @property
    def engine(self):
        return self._engine

    def _sample_planes(self, seed):
        N, K = self.num_envs, self.num_keywords
        base = int(np.random.SeedSequence().entropy % (2**62)) if seed is None else int(seed)
        if self.param_sampler == "vectorised":
            if not self._implicit:      # the default constructor's keyword set: sample_random_keywords' eight draws for all envs at once
                return synthetic.explicit_keyword_planes(N, K, base + self._env_id_base), base
            kc = self.keyword_config
            return synthetic.implicit_keyword_planes(N, K, base + self._env_id_base,
                                                     mean_volume=kc.get("mean_volume", 128),
                                                     cvr=kc.get("conversion_rate", 0.8),
                                                     no_vol_prob=kc.get("no_vol_prob", 0.0)), base
        planes = np.zeros((8, N, K), dtype=np.float32)
        for e in range(N):      # env e is seeded like gymnasium's vector envs: seed + (global) index
            rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(base + self._env_id_base + e)))
            if self._implicit:
                planes[:, e] = utils.implicit_params_to_planes(
                    utils.sample_implicit_keyword_params(K, rng, self.keyword_config))
            else:
                planes[:, e] = utils.explicit_params_to_planes(utils.sample_random_keyword_params(K, rng))
        return planes, base

    # ------------------------------------------------------------------ gymnasium.vector-style API
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        eng = self._ensure_engine(seed)
        if seed is not None or not self._have_keywords:
            if self.param_sampler == "device":
                base = int(np.random.SeedSequence().entropy % (2**62)) if seed is None else int(seed)
                seeds = (np.arange(self.num_envs, dtype=np.uint64) + np.uint64(base + self._env_id_base))
                eng.reset(seeds=seeds)
                if self._implicit:
                    kc = self.keyword_config
                    load = kc.get("load_quant_func")
                    eng.generate_keywords(load(kc), kc.get("no_vol_prob", 0.0))
                else:                   # the default constructor: sample_random_keywords' law (gymnasium_kw_utils.py:113-156)
                    eng.generate_explicit_keywords()
            else:
                planes, base = self._sample_planes(seed)
                eng.set_all_params(planes)
                seeds = (np.arange(self.num_envs, dtype=np.uint64) + np.uint64(base + self._env_id_base))
                eng.reset(seeds=seeds)
            self._have_keywords = True
        else:
            eng.reset()
        if options:
            self.max_days = int(options.get("max_days", self.max_days))
            self.loss_threshold = float(options.get("loss_threshold", self.loss_threshold))
            eng.set_limits(self.max_days, self.loss_threshold)
        return self._zero_obs(), {}

    def _zero_obs(self):
        N, K = self.num_envs, self.num_keywords
        obs = dict(impressions=np.zeros((N, K), np.int64), buyside_clicks=np.zeros((N, K), np.int64),
                   cost=np.zeros((N, K), np.float32), sellside_conversions=np.zeros((N, K), np.int64),
                   revenue=np.zeros((N, K), np.float32), cumulative_profit=np.zeros((N, 1), np.float32),
                   days_passed=np.zeros((N, 1), np.float32))
        return self._flatten(obs) if self.flat else obs

    @staticmethod
    def _flatten(obs):
        return np.concatenate([obs[k].reshape(obs[k].shape[0], -1).astype(np.float32) for k in utils.FLAT_OBS_KEYS], axis=1)

    def _split_actions(self, actions):
        N, K = self.num_envs, self.num_keywords
        if isinstance(actions, dict):
            bids = np.asarray(actions["keyword_bids"], dtype=np.float32).reshape(N, K)
            if "budget" in actions:
                self.budget = np.asarray(actions["budget"], dtype=np.float32).reshape(N)
        else:       # flat [N, K+1] = [budget, bids...] (sorted keys, flat_array.py:52,76)
            a = np.asarray(actions, dtype=np.float32).reshape(N, K + 1)
            self.budget = np.ascontiguousarray(a[:, 0])
            bids = np.ascontiguousarray(a[:, 1:])
        return bids, self.budget

    def action_buffers(self):
        """{"keyword_bids": [N, K], "budget": [N]} as views of the engine's page-locked action buffers: a policy that writes
        its actions into them and passes this dict to step() skips the staging copy (0.1 ms at 4096 x 256)"""
        assert self._engine is not None and not self.flat, "reset first; dict actions only"
        bids, budget = self._engine.action_buffers()
        return {"keyword_bids": bids, "budget": budget}

    def step(self, actions):
        assert self._have_keywords, "reset required, need to generate keywords to bid on"
        N, K = self.num_envs, self.num_keywords
        if self.flat:
            # one array each way; un/flattening happens on the device (adc_engine_step_flat)
            if isinstance(actions, dict):
                a = np.empty((N, K + 1), dtype=np.float32)
                a[:, 0] = np.asarray(actions.get("budget", self.budget), dtype=np.float32).reshape(N)
                a[:, 1:] = np.asarray(actions["keyword_bids"], dtype=np.float32).reshape(N, K)
                actions = a
            obs, reward, term, trunc = self._engine.step_flat(np.asarray(actions, dtype=np.float32).reshape(N, K + 1))
            term, trunc = term.astype(bool), trunc.astype(bool)
            reward = reward.copy() if self.copy else reward
            obs = obs.copy() if self.copy else obs
            infos = {}
            done = term | trunc
            if self.autoreset and done.any():
                # same-step autoreset (metadata["autoreset_mode"]): the terminal observation goes to infos["final_obs"], and
                # obs carries the first observation of the next episode - all zeros, as reset() returns it
                # (gymnasium_kw_env.py:340-342); the engine has already restarted those envs
                infos["_final_obs"] = done.copy()
                if done.all():                  # the usual case (all envs share max_days): one copy out, fresh zeros back
                    infos["final_obs"] = obs if self.copy else obs.copy()
                    obs = np.zeros_like(obs)
                else:
                    infos["final_obs"] = obs[done]
                    if not obs.flags.writeable or not self.copy:
                        obs = obs.copy()
                    obs[done] = 0.0
            return obs, reward, term, trunc, infos
        bids, budget = self._split_actions(actions)
        out = self._engine.step(bids, budget, copy=False)
        term, trunc = out["terminated"].astype(bool), out["truncated"].astype(bool)
        if self.copy:
            obs = dict(impressions=out["impressions"].astype(np.int64), buyside_clicks=out["buyside_clicks"].astype(np.int64),
                       cost=out["cost"].copy(), sellside_conversions=out["sellside_conversions"].astype(np.int64),
                       revenue=out["revenue"].copy(), cumulative_profit=out["cumulative_profit"].astype(np.float32)[:, None],
                       days_passed=out["days_passed"].astype(np.float32)[:, None])
            reward = out["reward"].copy()
        else:
            obs = dict(impressions=out["impressions"], buyside_clicks=out["buyside_clicks"], cost=out["cost"],
                       sellside_conversions=out["sellside_conversions"], revenue=out["revenue"],
                       cumulative_profit=out["cumulative_profit"].astype(np.float32)[:, None],
                       days_passed=out["days_passed"].astype(np.float32)[:, None])
            reward = out["reward"]
        infos = {}
        done = term | trunc
        if self.autoreset and done.any():
            # same-step autoreset: the engine already restarted those envs (day=0, cum=0, keywords kept, as
            # reset() without a seed does in the reference, gymnasium_kw_env.py:303,327-328)
            infos["_final_obs"] = done.copy()
            if done.all():                      # the usual case (all envs share max_days): one copy out, fresh zeros back
                infos["final_obs"] = obs if self.copy else {k: np.array(v) for k, v in obs.items()}
                obs = {k: np.zeros_like(v) for k, v in obs.items()}
            else:
                infos["final_obs"] = {k: v[done] for k, v in obs.items()}
                for k in obs:                   # obs of a finished env = the reset observation of its next episode: zeros
                    if not self.copy:
                        obs[k] = np.array(obs[k])
                    obs[k][done] = 0
        return obs, reward, term, trunc, infos

    # device-resident stepping for policies that live on the GPU (no PCIe on the step path)
    def step_device(self, d_flat_actions=None):
        if d_flat_actions is not None:
            if self._shards > 1:
                raise NotImplementedError("device-resident actions need one engine: construct with engine_shards=1")
            from ._ffi import check
            check(self._engine._lib.adc_engine_set_flat_actions_device(self._engine._h, d_flat_actions))
        self._engine.step_device()

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


class RLlibVectorEnvAdapter:
    """ray.rllib.env.VectorEnv call shapes (vector_reset / reset_at / vector_step / get_sub_environments)
    over one BiddingSimulationVectorEnv with flat observations - what `FlatArrayWrapper(bidding_sim_creator(cfg))`
    replicated num_envs times gives RLlib (adcraft/RL/train_agent.ipynb cell 10)."""

    def __init__(self, vec: BiddingSimulationVectorEnv):
        assert vec.flat, "construct the vector env with flat=True"
        self.vec = vec
        self.num_envs = vec.num_envs
        K = vec.num_keywords
        self.observation_space = _spaces.Box(low=-float("inf"), high=float("inf"), shape=(5 * K + 2,), dtype=np.float32)
        self.action_space = _spaces.Box(low=0.01, high=float("inf"), shape=(K + 1,), dtype=np.float32)

    def vector_reset(self, *, seeds=None, options=None):
        seed = None if not seeds or seeds[0] is None else int(seeds[0])
        obs, _ = self.vec.reset(seed=seed)
        return [o for o in obs], [{} for _ in range(self.num_envs)]

    def reset_at(self, index=None, *, seed=None, options=None):
        mask = np.zeros(self.num_envs, dtype=np.uint8)
        mask[index or 0] = 1
        self.vec.engine.reset(env_mask=mask)
        return np.zeros(self.observation_space.shape, dtype=np.float32), {}

    def vector_step(self, actions):
        obs, rew, term, trunc, _ = self.vec.step(np.stack([np.asarray(a, dtype=np.float32) for a in actions]))
        return [o for o in obs], rew.tolist(), term.tolist(), trunc.tolist(), [{} for _ in range(self.num_envs)]

    def get_sub_environments(self):
        return []


class SB3VecEnvAdapter:
    """stable_baselines3.common.vec_env.VecEnv call shapes (reset -> obs, step_async / step_wait ->
    (obs, rewards, dones, infos)) over one BiddingSimulationVectorEnv with flat observations."""

    def __init__(self, vec: BiddingSimulationVectorEnv):
        assert vec.flat and vec.autoreset
        self.vec = vec
        self.num_envs = vec.num_envs
        K = vec.num_keywords
        self.observation_space = _spaces.Box(low=-float("inf"), high=float("inf"), shape=(5 * K + 2,), dtype=np.float32)
        self.action_space = _spaces.Box(low=0.01, high=float("inf"), shape=(K + 1,), dtype=np.float32)
        self._actions = None
        self._seed = None

    def seed(self, seed=None):
        self._seed = seed
        return [None if seed is None else seed + i for i in range(self.num_envs)]

    def reset(self):
        obs, _ = self.vec.reset(seed=self._seed)
        self._seed = None
        return obs

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        obs, rew, term, trunc, infos = self.vec.step(self._actions)
        dones = term | trunc
        out_infos = [{"TimeLimit.truncated": bool(tr and not te)} for te, tr in zip(term, trunc)]
        if dones.any():
            final = infos["final_obs"]
            for j, i in enumerate(np.nonzero(dones)[0]):
                out_infos[i]["terminal_observation"] = final[j]
        return obs, rew.astype(np.float32), dones, out_infos      # (obs of finished envs is already the zero reset observation)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.vec.close()
