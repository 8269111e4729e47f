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
    queue: np.ndarray  # int32 [queue_size]
    queue_index: np.ndarray
    game_over: np.ndarray  # bool
    score: np.ndarray  # float32
    _token: object = dataclasses.field(default=None, repr=False, compare=False)

    def replace(self, **changes) -> "State":
        changes.setdefault("_token", None)  # an edited state is no longer the one resident on the device
        return dataclasses.replace(self, **changes)


# ---------------------------------------------------------------------------------------------------------------------
# The individual rule functions of reference functional/core.py:46-302 as host-side NumPy (arguments and returns as in the
# reference).  The engine evaluates the same rules inside its step kernel (csrc/tg_kernel_bits.h); these are for callers
# and tests that use the pieces on their own — e.g. the reference's tests/test_functional/test_core/*.
def _matrix(tetrominoes, active_tetromino, rotation):
    return np.asarray(tetrominoes.matrices[int(active_tetromino), int(rotation) & 3])


def create_board(config: EnvConfig, tetrominoes=None) -> np.ndarray:
    """An empty playfield with bedrock (1) below and on both sides, none on top (functional/core.py:46-63)."""
    p = config.padding
    board = np.ones((config.height + p, config.width + 2 * p), np.int8)
    board[: config.height, p: p + config.width] = 0
    return board


def get_initial_x_y(config: EnvConfig, tetrominoes=None, active_tetromino=0):
    """Spawn position: horizontally centred 4x4 box, top row (functional/core.py:66-83)."""
    return (config.width + 2 * config.padding) // 2 - 2, 0


def _window(board, x, y):
    """The 4x4 window of `board` at (x, y), clamped into the array the way lax.dynamic_slice clamps its start."""
    b = np.asarray(board)
    y0 = min(max(int(y), 0), b.shape[0] - 4)
    x0 = min(max(int(x), 0), b.shape[1] - 4)
    return b[y0: y0 + 4, x0: x0 + 4], x0, y0


def collision(board, tetromino, x, y) -> np.bool_:
    """Does the 4x4 matrix at (x, y) overlap a filled cell?  (functional/core.py:86-100)"""
    win, _, _ = _window(board, x, y)
    return np.bool_(((win > 0) & (np.asarray(tetromino) > 0)).any())


def project_tetromino(board, tetromino, x, y, tetromino_id) -> np.ndarray:
    """board + matrix * id at (x, y); no collision check (functional/core.py:103-121)."""
    out = np.array(board, copy=True)
    _, x0, y0 = _window(out, x, y)
    out[y0: y0 + 4, x0: x0 + 4] += (np.asarray(tetromino) * tetromino_id).astype(out.dtype)
    return out


def score(config: EnvConfig, rows_cleared) -> np.int32:
    """0 / 100 / 300 / 500 / 800 for 0..4 rows (functional/core.py:124-146)."""
    return np.int32((0, 100, 300, 500, 800)[int(rows_cleared)] if 0 <= int(rows_cleared) <= 4 else 0)


def graviy_step(tetrominoes, board, active_tetromino, rotation, x, y):
    """One row down if the piece is free below, else unchanged (the reference's spelling; functional/core.py:150-182)."""
    free = not collision(board, _matrix(tetrominoes, active_tetromino, rotation), x, int(y) + 1)
    return int(y) + 1 if free else int(y)


def clear_filled_rows(config: EnvConfig, tetrominoes, board):
    """→ (board, rows cleared).  Full playfield rows are removed, the rows above move down (functional/core.py:185-227).  The
    vacated top rows are copies of the OLD top row: the reference marks cleared rows with the gather index -height, which
    jnp.take wraps to row 0 (read from jax semantics, not pinned here — DESIGN.md §5); row 0 is empty in any reachable game."""
    p, H, W = config.padding, config.height, config.width
    field = np.asarray(board)[:H, p: p + W]
    full = (field > 0).all(axis=1)
    n = int(full.sum())
    if n == 0:
        return np.array(board, copy=True), np.int32(0)
    kept = field[~full]
    new_field = np.concatenate([np.repeat(field[:1], n, axis=0), kept], axis=0)
    out = np.array(board, copy=True)
    out[:H, p: p + W] = new_field
    return out, np.int32(n)


def hard_drop(board, tetromino, x, y):
    """→ (y where the piece comes to rest, reward 2 per row dropped)  (functional/core.py:230-251)."""
    ny = int(y)
    while not collision(board, tetromino, x, ny + 1):
        ny += 1
    return ny, np.int32(2 * (ny - int(y)))


def lock_active_tetromino(config: EnvConfig, tetrominoes, board, active_tetromino, rotation, x, y):
    """→ (board, reward, rows cleared): project the piece with its id, clear rows, score them (functional/core.py:254-281)."""
    placed = project_tetromino(board, _matrix(tetrominoes, active_tetromino, rotation), x, y, int(tetrominoes.ids[int(active_tetromino)]))
    cleared, n = clear_filled_rows(config, tetrominoes, placed)
    return cleared, score(config, n), n


def check_game_over(tetrominoes, board, active_tetromino, rotation, x, y) -> np.bool_:
    """A freshly spawned piece that already collides ends the game (functional/core.py:284-302)."""
    return collision(board, _matrix(tetrominoes, active_tetromino, rotation), x, y)
