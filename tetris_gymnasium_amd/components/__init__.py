"""The reference's component objects (tetromino, queue, holder, randomizer) as importable host-side classes."""
from .tetromino import Pixel, Tetromino  # noqa: F401
from .tetromino_holder import TetrominoHolder  # noqa: F401
from .tetromino_queue import TetrominoQueue  # noqa: F401
from .tetromino_randomizer import BagRandomizer, Randomizer, TrueRandomizer  # noqa: F401
