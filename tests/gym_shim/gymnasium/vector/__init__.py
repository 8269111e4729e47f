"""gymnasium.vector look-alike (see the package docstring): VectorEnv's documented attributes and the AutoresetMode enum of >= 1.1."""
from enum import Enum

from . import utils  # noqa: F401


class AutoresetMode(Enum):
    NEXT_STEP = "NextStep"
    SAME_STEP = "SameStep"
    DISABLED = "Disabled"


class VectorEnv:
    metadata: dict = {}
    spec = None
    render_mode = None
    closed = False
    observation_space = None
    action_space = None
    single_observation_space = None
    single_action_space = None
    num_envs: int

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, actions):
        raise NotImplementedError

    def close(self, **kwargs):
        self.closed = True

    def close_extras(self, **kwargs):
        pass

    @property
    def unwrapped(self):
        return self
