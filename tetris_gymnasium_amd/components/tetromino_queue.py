"""The preview queue (drop-in for reference components/tetromino_queue.py:7-66): the next `size` piece indices, refilled from a
randomizer so that it is always full.  Host-side object; the engine keeps the same queue per env on the device
(tg_env_rec.queue) and `Tetris.queue` mirrors it."""
from __future__ import annotations

from collections import deque

from .tetromino_randomizer import Randomizer


class TetrominoQueue:
    def __init__(self, randomizer: Randomizer, size=4):
        self.randomizer = randomizer
        self.size = size
        self.queue = deque(maxlen=size)

    def reset(self, seed=None):
        """Restart the randomizer (reseeding it when a seed is given) and draw `size` pieces."""
        self.randomizer.reset(seed)
        self.queue.clear()
        while len(self.queue) < self.size:
            self.queue.append(self.randomizer.get_next_tetromino())

    def get_next_tetromino(self):
        """Hand out the head and draw one piece behind the tail."""
        head = self.queue.popleft()
        self.queue.append(self.randomizer.get_next_tetromino())
        return head

    def get_queue(self):
        return list(self.queue)

    def copy(self, randomizer: Randomizer):
        """A queue with the same content on top of `randomizer` (pass a copy of this queue's randomizer)."""
        twin = TetrominoQueue(randomizer, self.size)
        twin.queue = deque(self.queue, maxlen=self.size)
        return twin
