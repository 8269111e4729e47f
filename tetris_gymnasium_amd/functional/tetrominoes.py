"""TETROMINOES LUT as NumPy arrays (mirror of reference functional/tetrominoes.py:124-165)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..components.tetromino import COLORS, NATIVE_MATRICES


@dataclass(frozen=True)
class Tetrominoes:
    base_pixels: np.ndarray
    base_pixel_colors: np.ndarray
    ids: np.ndarray
    colors: np.ndarray
    matrices: np.ndarray  # [7, 4, 4, 4] int8, binary, zero-padded bottom/right


def _build() -> Tetrominoes:
    mats = np.zeros((7, 4, 4, 4), np.int8)
    for p, nat in enumerate(NATIVE_MATRICES):
        for r in range(4):
            m = np.rot90(nat, k=r)
            mats[p, r, : m.shape[0], : m.shape[1]] = m
    return Tetrominoes(
        base_pixels=np.array([0, 1], np.int8),
        base_pixel_colors=np.array([[0, 0, 0], [128, 128, 128]], np.uint8),
        ids=np.arange(2, 9, dtype=np.int8),
        colors=np.array(COLORS, np.uint8),
        matrices=mats,
    )


TETROMINOES = _build()


def get_tetromino_matrix(tetrominoes: Tetrominoes, tetromino_id: int, rotation: int) -> np.ndarray:
    return tetrominoes.matrices[tetromino_id, rotation]


def get_tetromino_color(tetrominoes: Tetrominoes, tetromino_id: int) -> np.ndarray:
    """RGB colour of tetromino `tetromino_id` (reference functional/tetrominoes.py:169-180)."""
    return tetrominoes.colors[int(tetromino_id)]
