"""Stand-in modules so the UNMODIFIED reference NumPy env can be imported in the build container.

TEST INFRASTRUCTURE ONLY (golden-vector generation).  `gymnasium` and `cv2` are not installed here
and there is no network.  The reference's `Tetris.step` touches gymnasium only through
`action_space.contains` (reference envs/tetris.py:215) and `gym.Env.reset` seeding of an RNG the env
never reads (envs/tetris.py:287); cv2 only in `render`.  These stand-ins therefore contribute no
arithmetic to any golden vector.  They are this repo's own code, not reference code.

Usage (this container only; /root/reference does not exist on the GPU box):

    from oracle.refgen.standins import import_reference
    ref = import_reference()          # namespace with Tetris, GroupedActionsObservations, ...
"""
from __future__ import annotations

import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


class _Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype


class Box(_Space):
    def __init__(self, low=0, high=1, shape=None, dtype=np.float32):
        super().__init__(shape, dtype)
        self.low = low
        self.high = high


class Discrete(_Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)
        self._rng = np.random.default_rng(0)

    def contains(self, x) -> bool:
        if isinstance(x, (bool, np.bool_)):
            return False
        if isinstance(x, (int, np.integer)):
            return self.start <= int(x) < self.start + self.n
        if isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            return self.start <= int(x) < self.start + self.n
        return False

    def sample(self):
        return int(self._rng.integers(self.start, self.start + self.n))


class Dict(_Space):
    def __init__(self, spaces=None):
        super().__init__(None, None)
        self.spaces = dict(spaces or {})

    def keys(self):
        return self.spaces.keys()

    def __getitem__(self, k):
        return self.spaces[k]


class Env:
    metadata: dict = {}
    render_mode = None
    observation_space = None
    action_space = None

    def reset(self, *, seed=None, options=None):
        return None

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def action_space(self):
        return self._action_space if self._action_space is not None else self.env.action_space

    @action_space.setter
    def action_space(self, v):
        self._action_space = v

    @property
    def observation_space(self):
        return (
            self._observation_space
            if self._observation_space is not None
            else self.env.observation_space
        )

    @observation_space.setter
    def observation_space(self, v):
        self._observation_space = v

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()


class ObservationWrapper(Wrapper):
    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info

    def observation(self, observation):
        raise NotImplementedError


def _install() -> None:
    if "gymnasium" in sys.modules and getattr(sys.modules["gymnasium"], "__standin__", False):
        return
    gym = types.ModuleType("gymnasium")
    gym.__standin__ = True
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box, spaces.Discrete, spaces.Dict = Box, Discrete, Dict
    core = types.ModuleType("gymnasium.core")
    core.ActType = object
    core.RenderFrame = object
    core.ObsType = object
    utils = types.ModuleType("gymnasium.utils")
    seeding = types.ModuleType("gymnasium.utils.seeding")
    seeding.RandomNumberGenerator = np.random.Generator
    utils.seeding = seeding
    gym.Env, gym.Wrapper, gym.ObservationWrapper = Env, Wrapper, ObservationWrapper
    gym.spaces, gym.core, gym.utils = spaces, core, utils
    gym.register = lambda *a, **k: None
    sys.modules.update(
        {
            "gymnasium": gym,
            "gymnasium.spaces": spaces,
            "gymnasium.core": core,
            "gymnasium.utils": utils,
            "gymnasium.utils.seeding": seeding,
        }
    )
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")


def import_reference() -> types.SimpleNamespace:
    """Import the reference NumPy env (read-only mount) behind the stand-ins."""
    sys.dont_write_bytecode = True
    _install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from tetris_gymnasium.envs.tetris import Tetris
    from tetris_gymnasium.mappings.actions import ActionsMapping
    from tetris_gymnasium.mappings.rewards import RewardsMapping
    from tetris_gymnasium.wrappers.grouped import GroupedActionsObservations
    from tetris_gymnasium.wrappers.observation import FeatureVectorObservation, RgbObservation
    from tetris_gymnasium.components.tetromino import Tetromino

    return types.SimpleNamespace(
        Tetris=Tetris,
        ActionsMapping=ActionsMapping,
        RewardsMapping=RewardsMapping,
        GroupedActionsObservations=GroupedActionsObservations,
        FeatureVectorObservation=FeatureVectorObservation,
        RgbObservation=RgbObservation,
        Tetromino=Tetromino,
    )
