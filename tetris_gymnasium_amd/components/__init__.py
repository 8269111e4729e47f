"""Host-side views of the reference's component objects (tetromino, queue, holder, randomizer)."""
from .tetromino import Pixel, Tetromino  # noqa: F401
