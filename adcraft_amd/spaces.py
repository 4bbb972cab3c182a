"""Action / observation spaces (adcraft/gymnasium_kw_utils.py:31-64).

gymnasium is used when it is importable; otherwise a minimal Box/Dict with the same constructor,
`contains`, `sample` and `shape/dtype/low/high` attributes stands in, so the env works (and its
tests run) on boxes without gymnasium.
"""
import numpy as np

try:  # pragma: no cover - depends on the image
    from gymnasium.spaces import Box, Dict  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # ImportError or a broken install
    HAVE_GYMNASIUM = False

    class Space:
        def __init__(self):
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            super().__init__()
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=np.float64)
            self.high = np.full(self.shape, high, dtype=np.float64)
            if seed is not None:
                self.seed(seed)

        def contains(self, x):
            x = np.asarray(x)
            if x.shape != self.shape or not np.can_cast(x.dtype, self.dtype, casting="same_kind"):
                return False
            return bool(np.all(x >= self.low) and np.all(x <= self.high))

        __contains__ = contains

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e6)
            hi = np.where(np.isfinite(self.high), self.high, lo + 1.0 + self._rng.exponential(size=self.shape))
            v = self._rng.uniform(lo, np.maximum(hi, lo))
            if np.issubdtype(self.dtype, np.integer):
                v = np.floor(v)
            return v.astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(Space, dict):
        def __init__(self, spaces=None, seed=None, **kw):
            Space.__init__(self)
            dict.__init__(self, spaces or {}, **kw)
            self.spaces = self

        def contains(self, x):
            return isinstance(x, dict) and set(x.keys()) == set(self.keys()) and all(self[k].contains(x[k]) for k in self)

        def sample(self):
            return {k: s.sample() for k, s in self.items()}

        def seed(self, seed=None):
            for i, s in enumerate(self.values()):
                s.seed(None if seed is None else seed + i)
            return [seed]


def get_action_space(num_keywords):
    """adcraft/gymnasium_kw_utils.py:31-42"""
    return Dict({
        "keyword_bids": Box(low=0.01, high=float("Inf"), shape=(num_keywords,), dtype=np.float32),
        "budget": Box(low=0.01, high=float("Inf"), shape=(1,), dtype=np.float32),
    })


def get_observation_space(num_keywords, budget):
    """adcraft/gymnasium_kw_utils.py:45-64"""
    def nonneg_int():
        return Box(low=0, high=float("Inf"), shape=(num_keywords,), dtype=int)

    def nonneg_float():
        return Box(low=0, high=float("Inf"), shape=(num_keywords,), dtype=np.float32)
    return Dict({
        "impressions": nonneg_int(),
        "buyside_clicks": nonneg_int(),
        "cost": Box(low=0, high=budget, shape=(num_keywords,), dtype=np.float32),
        "sellside_conversions": nonneg_int(),
        "revenue": nonneg_float(),
        "cumulative_profit": Box(low=-float("Inf"), high=float("Inf"), shape=(1,), dtype=np.float32),
        "days_passed": Box(low=0, high=float("Inf"), shape=(1,), dtype=np.float32),
    })
