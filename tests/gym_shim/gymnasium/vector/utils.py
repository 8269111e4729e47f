"""gymnasium.vector.utils.batch_space look-alike: n copies of a space as one batched space (Discrete -> MultiDiscrete, Box -> Box with a
leading dimension, Dict -> Dict of batched spaces)."""
import numpy as np

from .. import spaces


def batch_space(space, n=1):
    if isinstance(space, spaces.Discrete):
        return spaces.MultiDiscrete(np.full((n,), space.n, np.int64), dtype=np.int64)
    if isinstance(space, spaces.Box):
        rep = (n,) + (1,) * len(space.shape)
        return spaces.Box(np.tile(space.low, rep), np.tile(space.high, rep), dtype=space.dtype)
    if isinstance(space, spaces.Dict):
        return spaces.Dict({k: batch_space(s, n) for k, s in space.items()})
    raise ValueError(f"cannot batch {type(space).__name__}")
