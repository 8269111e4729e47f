"""Functional surface (mirror of tetris_gymnasium/functional): EnvConfig, State, TETROMINOES."""
from .core import EnvConfig, State  # noqa: F401
from .tetrominoes import TETROMINOES, Tetrominoes, get_tetromino_color, get_tetromino_matrix  # noqa: F401
