"""Pixel / Tetromino data classes (reference components/tetromino.py:7-52) and the stock shape set."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Pixel:
    id: int
    color_rgb: list


@dataclass
class Tetromino(Pixel):
    matrix: np.ndarray

    def __copy__(self):
        return Tetromino(id=self.id, color_rgb=list(self.color_rgb), matrix=self.matrix.copy())


# native binary matrices and colours, order I O T S Z J L (reference envs/tetris.py:47-75)
NATIVE_MATRICES = (
    np.array([[0, 0, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]], np.uint8),
    np.array([[1, 1], [1, 1]], np.uint8),
    np.array([[0, 1, 0], [1, 1, 1], [0, 0, 0]], np.uint8),
    np.array([[0, 1, 1], [1, 1, 0], [0, 0, 0]], np.uint8),
    np.array([[1, 1, 0], [0, 1, 1], [0, 0, 0]], np.uint8),
    np.array([[1, 0, 0], [1, 1, 1], [0, 0, 0]], np.uint8),
    np.array([[0, 0, 1], [1, 1, 1], [0, 0, 0]], np.uint8),
)
COLORS = ([0, 240, 240], [240, 240, 0], [160, 0, 240], [0, 240, 0], [240, 0, 0], [0, 0, 240], [240, 160, 0])


def make_tetromino(index: int, rot: int = 0, val: int | None = None) -> Tetromino:
    """Tetromino `index` rotated `rot` times by np.rot90(k=1), cells valued `val` (default index + 2)."""
    val = index + 2 if val is None else val
    m = np.rot90(NATIVE_MATRICES[index] * np.uint8(val), k=rot % 4).copy()
    return Tetromino(index + 2, list(COLORS[index]), m)


def identify(t: Tetromino):
    """(index, rot, val) of a tetromino object, also for user-made ones (e.g. the reference tests'
    `vertical_i_tetromino` with binary cell values).  Raises for shapes outside the stock set."""
    m = np.asarray(t.matrix)
    binary = (m > 0).astype(np.uint8)
    vals = np.unique(m[m > 0])
    if vals.size != 1:
        raise ValueError("tetromino matrix must hold exactly one non-zero cell value")
    for index, nat in enumerate(NATIVE_MATRICES):
        if nat.shape != binary.shape:
            continue
        for rot in range(4):
            if np.array_equal(np.rot90(nat, k=rot), binary):
                return index, rot, int(vals[0])
    raise ValueError("custom tetromino shapes are not supported by the HIP engine (stock I O T S Z J L only)")
