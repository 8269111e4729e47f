"""The bag queue (7-bag by default; `EnvConfig.queue_size` = the bag length, 1..7) of the functional surface (reference functional/queue.py:20-67), on the host in NumPy.

`reset` / `step` / `batched_*` take these two callables as their queue plug points and evaluate the stock bag ON THE
DEVICE (the engine implements exactly this queue; any other callable is rejected there).  Called directly they are the
reference's functions: same arguments, same returns, NumPy arrays instead of jax arrays.  Piece sequences follow
`functional.prng` (parity unpinned against a live jax)."""
from __future__ import annotations

import numpy as np

from .prng import permutation, split


def create_bag_queue(config, key):
    """→ (queue int32[queue_size], queue_index 0): a fresh permutation of the piece ids (functional/queue.py:20-35)."""
    if not 1 <= int(config.queue_size) <= 7:  # (arange(queue_size) beyond 7 would index past the seven tetrominoes)
        raise NotImplementedError("queue_size (the bag length) must be 1..7")
    return permutation(np.asarray(key, np.uint32), config.queue_size), 0


def bag_queue_get_next_element(config, queue, queue_index, key):
    """→ (piece, queue, queue_index, key): the next piece of the bag; a used-up bag is replaced by a permutation drawn from
    split(key)[1] and the key by split(key)[0] (functional/queue.py:38-67)."""
    queue = np.asarray(queue, np.int32)
    if int(queue_index) >= config.queue_size:
        new_key, subkey = split(np.asarray(key, np.uint32))
        fresh, start = create_bag_queue(config, subkey)
        return np.int32(fresh[start]), fresh, start + 1, new_key
    return np.int32(queue[int(queue_index)]), queue, int(queue_index) + 1, np.asarray(key, np.uint32)
