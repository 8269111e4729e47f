"""A gymnasium LOOK-ALIKE for tests/test_gym_branches.py — TEST INFRASTRUCTURE, not gymnasium.

Neither image of this project has gymnasium, so the `HAVE_GYMNASIUM` branches of tetris_gymnasium_amd (`_gym_compat.py`, `vector.py`,
`envs/__init__.py`) had never executed anywhere.  This package restates, from gymnasium 1.1's documented public API, exactly the names
those branches touch — `Env`, `Wrapper`, `spaces`, `register` / `make`, `vector.VectorEnv`, `vector.AutoresetMode`,
`vector.utils.batch_space` — so that they RUN (import errors, attribute errors, wrong argument shapes show up).  It proves nothing
about conformance with the real package: tests/test_gymnasium_conformance.py does that wherever gymnasium imports."""
from . import spaces, vector  # noqa: F401

__version__ = "1.1.1-shim"
_registry = {}


class Env:
    metadata: dict = {"render_modes": []}
    render_mode = None
    spec = None
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

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()


class ObservationWrapper(Wrapper):
    pass


def register(id, entry_point=None, **kwargs):  # noqa: A002 - gymnasium's own parameter name
    if id in _registry:
        raise ValueError(f"{id} is already registered")
    _registry[id] = (entry_point, kwargs)


def make(id, **kwargs):  # noqa: A002
    import importlib

    entry_point, defaults = _registry[id]
    if isinstance(entry_point, str):
        mod, _, attr = entry_point.partition(":")
        entry_point = getattr(importlib.import_module(mod), attr)
    return entry_point(**{**defaults.get("kwargs", {}), **kwargs})
