"""`RgbObservation`, `FeatureVectorObservation`: reference tetris_gymnasium/wrappers/observation.py:11-278."""
from __future__ import annotations

import numpy as np

from .. import _native as N
from .._gym_compat import spaces
from ._device import WrapperBase


class _ObservationWrapper(WrapperBase):
    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info


class RgbObservation(_ObservationWrapper):
    """Board on the left, queue strip top right, holder bottom right, as one RGB image (reference
    wrappers/observation.py:11-74).  Packed on the device from the wrapped env's own state (tg_observe_as, surface TG_OBS_RGB)."""

    def __init__(self, env):
        super().__init__(env)
        u = env.unwrapped
        self.observation_space = spaces.Box(
            low=0, high=255,
            shape=(u.height_padded, u.width_padded + max(u.holder.size, u.queue.size) * u.padding, 3), dtype=np.uint8)

    def observation(self, observation=None):
        u = self.env.unwrapped
        u._flush()  # (host-side edits of the env's attributes → device)
        return u._engine.observe_as(N.OBS_RGB)[0]


class FeatureVectorObservation(_ObservationWrapper):
    """Column heights, max height, holes, bumpiness as uint8 (reference wrappers/observation.py:118-278, incl. its
    row-wipe quirk: `board[active_tetromino_mask] = 0` indexes ROWS with the uint8 mask).  Computed on bit masks on
    the device from the wrapped env's own state (tg_features_as)."""

    def __init__(self, env, report_height=True, report_max_height=True, report_holes=True, report_bumpiness=True):
        super().__init__(env)
        u = env.unwrapped
        self.report_height, self.report_max_height = report_height, report_max_height
        self.report_holes, self.report_bumpiness = report_holes, report_bumpiness
        n = (u.width if report_height else 0) + int(report_max_height) + int(report_holes) + int(report_bumpiness)
        self.observation_space = spaces.Box(low=0, high=len(u.tetrominoes), shape=(n,), dtype=np.uint8)

    def select(self, feats):
        """Pick the reported entries out of [..., W + 3] vectors (heights | max | holes | bumpiness)."""
        W = self.unwrapped.width
        idx = (list(range(W)) if self.report_height else []) + ([W] if self.report_max_height else []) + \
              ([W + 1] if self.report_holes else []) + ([W + 2] if self.report_bumpiness else [])
        return np.ascontiguousarray(feats[..., idx])

    def observation(self, observation=None):
        u = self.env.unwrapped
        u._flush()
        return self.select(u._engine.features_as(placement=False, current=True)[1][0])
