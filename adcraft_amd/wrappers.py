"""FlatArrayWrapper - mirror of adcraft/wrappers/flat_array.py: Dict observation/action <-> flat Box."""
import numpy as np

from . import spaces as _spaces
from .gymnasium_kw_utils import flatten_dict_array


class FlatArrayWrapper:
    def __init__(self, env):
        self.env = env
        K = env.num_keywords
        # spaces.flatten_space of the Dict spaces: keys in sorted order
        self.observation_space = _spaces.Box(low=-float("inf"), high=float("inf"), shape=(5 * K + 2,), dtype=np.float32)
        self.action_space = _spaces.Box(low=0.01, high=float("inf"), shape=(K + 1,), dtype=np.float32)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def observation(self, observation):
        return flatten_dict_array(observation).astype(np.float32)

    def action(self, action):
        """unflatten: [budget, keyword_bids...] (sorted keys)"""
        a = np.asarray(action, dtype=np.float32)
        return {"budget": a[:1], "keyword_bids": a[1:]}

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(self.action(action))
        return flatten_dict_array(obs), reward, terminated, truncated, info      # flat_array.py:74-80

    def reset(self, *args, seed=None, options=None):
        obs, info = self.env.reset(*args, seed=seed, options=options)
        return self.observation(obs), info
