"""Tetromino randomizers (drop-in for reference components/tetromino_randomizer.py:8-142): which piece comes next.

Host-side NumPy objects with the reference's interface.  The engine runs the bag randomizer ON THE DEVICE for every env
of a batch (numpy's PCG64 + Generator.shuffle restated in csrc/, pinned against numpy by tests/golden/pcg64_streams.npz);
these classes are for code that uses a randomizer on its own, and `Tetris.randomizer` is a `BagRandomizer` whose fields
mirror the device-side generator of that env."""
from __future__ import annotations

import numpy as np


class Randomizer:
    """Base class: `get_next_tetromino()` returns the INDEX of the next piece, `reset(seed)` (re)starts the sequence."""

    def __init__(self, size: int):
        self.size = size
        self.rng: "np.random.Generator | None" = None

    def get_next_tetromino(self) -> int:
        raise NotImplementedError

    def reset(self, seed=None):
        """Gymnasium's pattern: a seed arrives with the first reset only; later resets keep the running generator.
        `seed` of None or 0 does not reseed (reference :40: `if seed and seed > 0`)."""
        if seed and seed > 0:
            self.rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        elif self.rng is None:
            self.rng = np.random.default_rng()


class BagRandomizer(Randomizer):
    """The 7-bag: every piece once per bag, the bag is reshuffled IN PLACE as soon as its last piece has been handed out
    (reference :49-102).  The stream is therefore a concatenation of successive in-place shuffles starting from arange."""

    def __init__(self, size):
        super().__init__(size)
        self.bag = np.arange(self.size, dtype=np.int8)
        self.index = 0

    def get_next_tetromino(self) -> int:
        piece = self.bag[self.index]
        self.index += 1
        if self.index >= len(self.bag):
            self.shuffle_bag()
        return piece

    def shuffle_bag(self):
        self.rng.shuffle(self.bag)
        self.index = 0

    def reset(self, seed=None):
        super().reset(seed)
        self.bag = np.arange(self.size, dtype=np.int8)
        self.shuffle_bag()

    def __copy__(self):
        twin = BagRandomizer(self.size)
        if self.rng is not None:  # (a generator restored from its state: cheaper than a deep copy)
            twin.rng = np.random.Generator(np.random.PCG64())
            twin.rng.bit_generator.state = self.rng.bit_generator.state
        twin.bag = self.bag.copy()
        twin.index = self.index
        return twin


class TrueRandomizer(Randomizer):
    """Independent uniform draws (reference :105-142); repeats are possible.  Not implemented by the engine
    (DESIGN.md §8): host-side only."""

    def get_next_tetromino(self) -> int:
        return int(self.rng.integers(0, self.size))

    def reset(self, seed=None):
        super().reset(seed)

    def __copy__(self):
        twin = TrueRandomizer(self.size)
        if self.rng is not None:
            twin.rng = np.random.Generator(np.random.PCG64())
            twin.rng.bit_generator.state = self.rng.bit_generator.state
        return twin
