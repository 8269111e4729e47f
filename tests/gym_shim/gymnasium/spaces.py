"""gymnasium.spaces look-alike (see the package docstring)."""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape, self.dtype = (None if shape is None else tuple(shape)), (None if dtype is None else np.dtype(dtype))
        self._rng = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n, self.start = int(n), int(start)

    def contains(self, x):
        if isinstance(x, (bool, np.bool_)):
            return False
        if isinstance(x, (int, np.integer)) or (isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer)):
            return self.start <= int(x) < self.start + self.n
        return False

    def sample(self, mask=None):
        return np.int64(self._rng.integers(self.start, self.start + self.n))


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None, start=None):
        self.nvec = np.asarray(nvec, dtype)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self, mask=None):
        return self._rng.integers(0, self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0)) and bool(np.all(x < self.nvec))


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        shape = tuple(shape) if shape is not None else np.shape(low)
        super().__init__(shape, dtype, seed)
        self.low = np.broadcast_to(np.asarray(low, dtype), shape)
        self.high = np.broadcast_to(np.asarray(high, dtype), shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.can_cast(x.dtype, self.dtype, "same_kind") and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def sample(self, mask=None):
        return self._rng.integers(self.low.astype(np.int64), self.high.astype(np.int64) + 1).astype(self.dtype)


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__(None, None, seed)
        self.spaces = dict(spaces or {}, **kw)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def __getitem__(self, k):
        return self.spaces[k]

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(self.spaces[k].contains(v) for k, v in x.items())

    def sample(self, mask=None):
        return {k: s.sample() for k, s in self.spaces.items()}
