"""EnvConfig / State of the functional surface (mirror of reference functional/core.py:11-42)."""
from __future__ import annotations

import dataclasses
from typing import NamedTuple

import numpy as np


class EnvConfig(NamedTuple):
    width: int
    height: int
    padding: int
    queue_size: int
    gravity_enabled: bool = True


@dataclasses.dataclass
class State:
    """Same fields / shapes as the reference's chex dataclass; leaves are NumPy arrays (a leading batch
    dimension on every leaf for the batched functions)."""

    rng_key: np.ndarray  # uint32 [2]
    board: np.ndarray  # int8 [H+p, W+2p]
    active_tetromino: np.ndarray  # int32
    rotation: np.ndarray
    x: np.ndarray
    y: np.ndarray
    queue: np.ndarray  # int32 [7]
    queue_index: np.ndarray
    game_over: np.ndarray  # bool
    score: np.ndarray  # float32
    _token: object = dataclasses.field(default=None, repr=False, compare=False)

    def replace(self, **changes) -> "State":
        changes.setdefault("_token", None)  # an edited state is no longer the one resident on the device
        return dataclasses.replace(self, **changes)


def create_board(config: EnvConfig, tetrominoes=None) -> np.ndarray:
    """functional/core.py:46-63: zeros H x W padded with bedrock (1) bottom/left/right."""
    b = np.zeros((config.height, config.width), np.int8)
    return np.pad(b, ((0, config.padding), (config.padding, config.padding)), constant_values=1)


def get_initial_x_y(config: EnvConfig, tetrominoes=None, active_tetromino=0):
    """functional/core.py:66-83."""
    return (config.width + config.padding * 2) // 2 - 4 // 2, 0
