"""Host-side jax.random pieces the functional surface needs (threefry2x32, PRNGKey, split, permutation of up to 7), restated from
jax 0.5.3 `_src/prng.py` / `_src/random.py` (partitionable threefry = the jax >= 0.5.0 default).  threefry2x32 itself is
pinned by the Random123 known-answer vectors (tests/test_oracle_rng.py); split / permutation are PARITY UNPINNED against a
live jax (none installable in the build container) — the device code and the oracle use the same restatement."""
from __future__ import annotations

import numpy as np

_M32 = 0xFFFFFFFF


def PRNGKey(seed: int) -> np.ndarray:
    """jax.random.PRNGKey for the threefry implementation: [hi32(seed), lo32(seed)]."""
    seed = int(seed)
    return np.array([(seed >> 32) & _M32, seed & _M32], np.uint32)


def _rotl(x, d):
    return ((x << d) | (x >> (32 - d))) & _M32


def threefry2x32(key, ctr):
    """Threefry-2x32, 20 rounds (Random123); pinned by the published known-answer vectors in tests."""
    k0, k1 = int(key[0]), int(key[1])
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0, x1 = (int(ctr[0]) + ks[0]) & _M32, (int(ctr[1]) + ks[1]) & _M32
    rots = ((13, 15, 26, 6), (17, 29, 16, 24))
    for blk in range(5):
        for r in rots[blk & 1]:
            x0 = (x0 + x1) & _M32
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(blk + 1) % 3]) & _M32
        x1 = (x1 + ks[(blk + 2) % 3] + blk + 1) & _M32
    return x0, x1


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split (threefry_partitionable): key_i = threefry(key, (0, i))."""
    return np.array([threefry2x32(key, (0, i)) for i in range(num)], np.uint32)


def threefry2x32_batch(k0, k1, c0, c1):
    """threefry2x32 on uint32 arrays (NumPy wrap-around arithmetic): the batched form of `threefry2x32` above."""
    k0, k1 = np.asarray(k0, np.uint32), np.asarray(k1, np.uint32)
    ks = (k0, k1, k0 ^ k1 ^ np.uint32(0x1BD11BDA))
    x0, x1 = (np.asarray(c0, np.uint32) + ks[0]).astype(np.uint32), (np.asarray(c1, np.uint32) + ks[1]).astype(np.uint32)
    rots = ((13, 15, 26, 6), (17, 29, 16, 24))
    for blk in range(5):
        for r in rots[blk & 1]:
            x0 = x0 + x1
            x1 = ((x1 << np.uint32(r)) | (x1 >> np.uint32(32 - r))) ^ x0
        x0 = x0 + ks[(blk + 1) % 3]
        x1 = x1 + ks[(blk + 2) % 3] + np.uint32(blk + 1)
    return x0, x1


def split_batch(keys, index: int = 0) -> np.ndarray:
    """split(key)[index] for every row of keys[B, 2] at once (batched_reset returns split(key)[0] per env: a Python loop over
    65 536 keys took a second)."""
    keys = np.asarray(keys, np.uint32).reshape(-1, 2)
    z = np.zeros(keys.shape[0], np.uint32)
    x0, x1 = threefry2x32_batch(keys[:, 0], keys[:, 1], z, z + np.uint32(index))
    return np.stack([x0, x1], axis=1)


def permutation(key, n: int) -> np.ndarray:
    """jax.random.permutation(key, arange(n)) for 1 <= n <= 7 (jax `_shuffle`: one round for 2 <= n, none for n = 1): key', sub =
    split(key); sort keys = 32 random bits per element from `sub` (threefry(sub, (0, i)).x ^ .y); stable argsort."""
    n = int(n)
    if n < 2:
        return np.arange(n, dtype=np.int32)
    sub = split(key)[1]
    x0, x1 = threefry2x32_batch(np.full(n, sub[0], np.uint32), np.full(n, sub[1], np.uint32), np.zeros(n, np.uint32),
                                np.arange(n, dtype=np.uint32))
    return np.argsort(x0 ^ x1, kind="stable").astype(np.int32)


def permutation7(key) -> np.ndarray:
    return permutation(key, 7)
