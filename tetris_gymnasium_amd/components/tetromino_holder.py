"""The hold slot(s) (drop-in for reference components/tetromino_holder.py:8-63).  Host-side object; the engine keeps the
held piece per env on the device (tg_env_rec.holder_*) and `Tetris.holder` mirrors it."""
from __future__ import annotations

from collections import deque
from typing import Optional

from .tetromino import Tetromino


class TetrominoHolder:
    def __init__(self, size=1):
        self.size = size
        self.queue = deque(maxlen=size)

    def swap(self, tetromino: Tetromino) -> Optional[Tetromino]:
        """Store `tetromino`; once the holder is full the oldest stored piece comes back, before that None."""
        out = self.queue.popleft() if len(self.queue) >= self.size else None
        self.queue.append(tetromino)
        return out

    def reset(self):
        self.queue.clear()

    def get_tetrominoes(self):
        return list(self.queue)

    def __copy__(self):
        twin = TetrominoHolder(self.size)
        twin.queue = deque(self.queue, maxlen=self.size)
        return twin
