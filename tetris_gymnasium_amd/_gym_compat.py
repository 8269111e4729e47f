"""gymnasium is an optional dependency: use it when importable (the adapter then IS a gymnasium.Env /
VectorEnv), otherwise degrade to duck-typed minimal spaces so the engine stays usable without it."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as gym
    from gymnasium import spaces

    HAVE_GYMNASIUM = True
    EnvBase = gym.Env
    try:
        from gymnasium.vector import VectorEnv as VectorEnvBase
    except Exception:  # very old gymnasium
        VectorEnvBase = object
except Exception:
    HAVE_GYMNASIUM = False
    gym = None

    class _Space:
        def __init__(self, shape=None, dtype=None):
            self.shape, self.dtype = shape, dtype
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

    class Discrete(_Space):
        def __init__(self, n, start=0):
            super().__init__((), np.int64)
            self.n, self.start = int(n), int(start)

        def contains(self, x):
            if isinstance(x, (bool, np.bool_)):
                return False
            if isinstance(x, (int, np.integer)) or (isinstance(x, np.ndarray) and x.shape == ()
                                                    and np.issubdtype(x.dtype, np.integer)):
                return self.start <= int(x) < self.start + self.n
            return False

        def sample(self):
            return int(self._rng.integers(self.start, self.start + self.n))

    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, np.int64)
            super().__init__(self.nvec.shape, np.int64)

        def sample(self):
            return self._rng.integers(0, self.nvec)

    class Box(_Space):
        def __init__(self, low=0, high=1, shape=None, dtype=np.float32):
            super().__init__(tuple(shape) if shape is not None else None, dtype)
            self.low, self.high = low, high

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    class Dict(_Space):
        def __init__(self, spaces_=None):
            super().__init__(None, None)
            self.spaces = dict(spaces_ or {})

        def keys(self):
            return self.spaces.keys()

        def __getitem__(self, k):
            return self.spaces[k]

        def contains(self, x):
            return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(
                self.spaces[k].contains(v) for k, v in x.items())

    class spaces:  # noqa: N801 - mimics the gymnasium.spaces module
        pass

    spaces.Discrete, spaces.MultiDiscrete, spaces.Box, spaces.Dict = Discrete, MultiDiscrete, Box, Dict

    class EnvBase:
        metadata: dict = {}
        render_mode = None

        def reset(self, *, seed=None, options=None):
            return None

        def close(self):
            pass

        @property
        def unwrapped(self):
            return self

    VectorEnvBase = object
