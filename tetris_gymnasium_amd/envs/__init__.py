"""Environment surfaces (mirror of tetris_gymnasium/envs)."""
from .tetris import Tetris  # noqa: F401
