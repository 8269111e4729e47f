"""`GroupedActionsObservations`: reference tetris_gymnasium/wrappers/grouped.py:17-294."""
from __future__ import annotations

import numpy as np

from .. import _native as N
from .._gym_compat import spaces
from ..envs.tetris import TetrisState
from ._device import WrapperBase, aux_engine, mirror
from .observation import FeatureVectorObservation, RgbObservation


class GroupedActionsObservations(WrapperBase):
    """action = column * 4 + rotations → teleport + hard drop; observation = one entry per action (the board each
    placement would produce, its feature vector when `observation_wrappers=[FeatureVectorObservation(env)]`, its RGB image when
    `observation_wrappers=[RgbObservation(env)]`); `info["action_mask"]` marks placements that collide with the frame.
    Enumeration, mask, features and the grouped transition all run on the device.  The RGB form (reference wrappers/grouped.py:
    181-204 feeding wrappers/observation.py:38-74 with every placement board and the env's holder / queue) is composed on the host:
    the placement boards of tg_grouped_enumerate through the colour table, beside the strip columns of the env's own RGB frame
    (tg_observe_as) — 4W small images per call of a single-env class are not worth a kernel.  Any other wrapper list is refused."""

    def __init__(self, env, observation_wrappers=None, terminate_on_illegal_action: bool = True):
        super().__init__(env)
        u = env.unwrapped
        if observation_wrappers and not (len(observation_wrappers) == 1 and
                                         isinstance(observation_wrappers[0], (FeatureVectorObservation, RgbObservation))):
            raise NotImplementedError("observation_wrappers: [FeatureVectorObservation(env)] and [RgbObservation(env)] are supported")
        self.observation_wrappers = observation_wrappers
        self._rgb = bool(observation_wrappers) and isinstance(observation_wrappers[0], RgbObservation)
        self.terminate_on_illegal_action = terminate_on_illegal_action
        self.action_space = spaces.Discrete(u.width * 4)
        single = observation_wrappers[-1].observation_space.shape if observation_wrappers else \
            (u.height_padded, u.width_padded)
        self.observation_space = spaces.Box(low=0, high=u.height * u.width, shape=(u.width * 4,) + tuple(single),
                                            dtype=np.float32)
        self.legal_actions_mask = np.ones(self.action_space.n)
        self._aux = aux_engine(env, obs_mode=N.OBS_NONE, grouped=True, features=bool(observation_wrappers) and not self._rgb,
                               terminate_on_illegal=terminate_on_illegal_action)
        self._seen = None  # the state version of the wrapped env that the auxiliary engine holds (wrappers/_device.py mirror)
        self._wb = self._writeback  # (ONE bound-method object: the env keeps it while our write-back is pending, Tetris._defer)

    def encode_action(self, x, r):
        return x * 4 + r

    def decode_action(self, action):
        return action // 4, action % 4

    # -- observation of the env's CURRENT state (the aux engine must already mirror it) ---------------
    def _observe(self):
        if self._rgb:
            u = self.env.unwrapped
            mask, boards = self._aux.placement_boards()
            self.legal_actions_mask = mask[0].astype(np.float64)
            frame = self._aux.observe_as(N.OBS_RGB)[0]  # the env's own image: its strip columns are every placement's
            colors = np.array([p.color_rgb for p in u.pixels], np.uint8)
            out = np.empty((boards.shape[1],) + frame.shape, np.uint8)
            out[:, :, :u.width_padded] = colors[boards[0]]
            out[:, :, u.width_padded:] = frame[None, :, u.width_padded:]
            return out, frame
        if self.observation_wrappers:
            placement, current = self._aux.features()
            self.legal_actions_mask = self._aux.legal_mask()[0].astype(np.float64)
            fv = self.observation_wrappers[0]
            return fv.select(placement[0]), fv.select(current[0])
        mask, boards = self._aux.placement_boards()
        self.legal_actions_mask = mask[0].astype(np.float64)
        return boards[0], None

    def _writeback(self):
        """the auxiliary engine's state → the wrapped env (runs when somebody asks the env: Tetris._settle)"""
        u = self.env.unwrapped
        b, r, g = self._aux.get_state()
        u.set_state(TetrisState(b, r, g))
        self._seen = u._version  # (the auxiliary engine holds exactly what was written back)

    def _mirror(self):
        """make the auxiliary engine hold the env's state; nothing to do while our own write-back is still pending (nobody has
        read or changed the env since this wrapper's last step: the auxiliary engine IS its state)"""
        if self.env.unwrapped._deferred is not self._wb:
            self._seen = mirror(self.env, self._aux, self._seen)

    def observation(self, observation=None):
        self._mirror()
        return self._observe()[0]

    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        self._mirror()
        grouped, current = self._observe()
        info["board"] = current if self.observation_wrappers else obs
        info["action_mask"] = self.legal_actions_mask
        return grouped, info

    def step(self, action):
        u = self.env.unwrapped
        self._mirror()
        if self.legal_actions_mask[action] == 0 and self.terminate_on_illegal_action:
            observation = np.ones(self.observation_space.shape) * self.observation_space.high
            return observation, u.rewards.invalid_action, True, False, \
                {"action_mask": self.legal_actions_mask, "lines_cleared": 0}
        out = self._aux.step_fetch_out([int(action)])  # grouped transition on the device
        term, lines, fl = out["terminated"], out["lines"], int(out["flags"][0])
        self._seen = None
        u._defer(self._wb)  # the env's state is the auxiliary engine's now; it is written back when somebody asks the env
        illegal = bool(fl & N.OUT_ILLEGAL)
        reward = 0  # (as the reference's Python computes it: wrappers/grouped.py:238, envs/tetris.py:461-474)
        if illegal:
            reward = u.rewards.invalid_action
        elif fl & N.OUT_COMMITTED:
            reward = u.rewards.game_over if term[0] else u.score(int(lines[0])) + u.rewards.alife
        grouped, current = self._observe()
        info = {"lines_cleared": int(lines[0]), "action_mask": self.legal_actions_mask}
        if not illegal:
            # (the base env's observation after the transition, wrappers/grouped.py:261-264 — packed from the auxiliary engine's state)
            info["board"] = current if self.observation_wrappers else {k: v[0] for k, v in self._aux.observe_as(N.OBS_DICT).items()}
        return grouped, reward, bool(term[0]), False, info
