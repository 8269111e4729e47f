"""Shared plumbing of the wrapper classes: a one-env auxiliary engine that mirrors the wrapped env's state on demand
and runs the wrapper's kernel (RGB pack, feature vectors, placement enumeration, grouped step) on it."""
from __future__ import annotations

from .. import _native as N
from ..engine import Engine


class WrapperBase:
    """Minimal gymnasium.Wrapper look-alike (works with and without gymnasium installed): forwards unknown
    attributes to the wrapped env, exposes `.env` and `.unwrapped`."""

    def __init__(self, env):
        self.env = env

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def close(self):
        u = self.unwrapped
        if getattr(u, "_deferred", None) is not None and self.__dict__.get("_aux") is not None:
            u._settle()  # (a pending write-back reads the auxiliary engine that is closed below)
        for a in ("_aux",):
            e = self.__dict__.get(a)
            if e is not None:
                e.close()
                self.__dict__[a] = None
        self.env.close()


def aux_engine(env, **kw) -> Engine:
    u = env.unwrapped
    return Engine(1, u.width, u.height, ruleset=N.RULESET_NUMPY, gravity=u.gravity_enabled, queue_size=u.queue.size,
                  device=u._engine.cfg.device, actions_mapping=u.actions, rewards_mapping=u.rewards, **kw)


def mirror(env, aux: Engine, seen=None) -> int:
    """Copy the wrapped env's full state (board, record, generator) into the auxiliary engine and return the env's state
    version (`Tetris._version`: every change of its device state counts) that the auxiliary engine now holds.  `seen`: the
    version it already holds — nothing is copied when that is still the env's state (a wrapper stepping in a loop would
    otherwise spend a third of its time copying a state back and forth that nobody touched)."""
    u = env.unwrapped
    u._flush()  # (host-side edits of the env's attributes count as a change)
    if seen is not None and seen == u._version:
        return seen
    st = u.get_state()
    aux.set_state(board=st.board, rec=st.rec, rng=st.rng)
    return u._version
