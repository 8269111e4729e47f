"""Functional `reset / step / batched_reset / batched_step` with the reference's signatures
(tetris_gymnasium/envs/tetris_fn.py:276-467), evaluated by the HIP engine (rule-set FUNCTIONAL).

Pure-functional contract kept: inputs are never mutated, a new `State` is returned.  A returned State
carries a private token naming the engine generation that produced it; passing it straight back (the
usual rollout loop) skips the host→device upload.  Stock TETROMINOES and the stock 7-bag queue only —
anything else is rejected explicitly.

PARITY UNPINNED (piece sequences): they follow jax 0.5.3's threefry split / permutation as restated from its source
(partitionable scheme) and the `jnp.take(-H)` row-0 corner of clear_filled_rows as read from jax semantics; neither is
pinned against a live jax (none installable in the build container).  `python -m oracle.refgen.make_golden_fn` pins
them wherever `import jax` works (writes tests/golden/fn_streams.npz, replayed by tests/test_oracle_fn_golden.py and
tests/test_gpu_fn_golden.py); until that file exists, engine == oracle is all that is verified.  Game logic, rewards and
observations are pinned by the reference's own known answers (tests/test_oracle_functional.py).
"""
from __future__ import annotations

import itertools

import numpy as np

from .. import _native as N
from ..engine import Engine
from ..functional.core import EnvConfig, State
from ..functional.queue import bag_queue_get_next_element, create_bag_queue
from ..functional.tetrominoes import TETROMINOES, Tetrominoes

ACTION_ID_TO_NAME = {
    0: "move_left", 1: "move_right", 2: "move_down", 3: "rotate_counterclockwise", 4: "rotate_clockwise",
    5: "do_nothing", 6: "hard_drop",
}

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


# ---------------------------------------------------------------------------------------------------
_engines: dict = {}
_ids = itertools.count(1)


class _Slot:
    def __init__(self, config: EnvConfig, batch: int):
        self.engine = Engine(batch, config.width, config.height, ruleset=N.RULESET_FUNCTIONAL,
                             gravity=config.gravity_enabled, queue_size=7, obs_mode=N.OBS_FN,
                             padding=config.padding)
        self.id = next(_ids)
        self.generation = 0


def _check(tetrominoes, config, *fns):
    if not isinstance(config, EnvConfig) and not (isinstance(config, tuple) and len(config) >= 4):
        raise TypeError("config must be an EnvConfig")
    if tetrominoes is not TETROMINOES:
        if not isinstance(tetrominoes, Tetrominoes) or not np.array_equal(tetrominoes.matrices, TETROMINOES.matrices) \
                or not np.array_equal(tetrominoes.ids, TETROMINOES.ids):
            raise NotImplementedError("the HIP engine implements the stock TETROMINOES table only")
    for f in fns:
        if f not in (create_bag_queue, bag_queue_get_next_element):
            raise NotImplementedError("the HIP engine implements the stock 7-bag queue only "
                                      "(create_bag_queue / bag_queue_get_next_element)")
    if config.padding != 4 or config.queue_size != 7:
        raise NotImplementedError("the HIP engine needs padding=4 and queue_size=7 (stock tetromino set)")


def _slot(config: EnvConfig, batch: int) -> _Slot:
    key = (tuple(config), batch)
    s = _engines.get(key)
    if s is None:
        s = _engines[key] = _Slot(EnvConfig(*config), batch)
    return s


def _pad(board_u8, config):
    p = config.padding
    return np.pad(board_u8.astype(np.int8), ((0, 0), (0, p), (p, p)), constant_values=1)


def _download(slot: _Slot, config) -> dict:
    board, rec, rng = slot.engine.get_state()
    return dict(
        rng_key=rng["key"].copy(), board=_pad(board, config), active_tetromino=rec["piece"].astype(np.int32),
        rotation=rec["rot"].astype(np.int32), x=rec["x"].astype(np.int32), y=rec["y"].astype(np.int32),
        queue=rec["bag"].astype(np.int32), queue_index=rec["bag_index"].astype(np.int32),
        game_over=(rec["flags"] & N.F_GAME_OVER).astype(bool), score=rec["score"].astype(np.float32))


def _upload(slot: _Slot, config, leaves: dict):
    e = slot.engine
    B, p = e.B, config.padding
    rec = np.zeros(B, N.ENV_REC)
    rec["piece"] = leaves["active_tetromino"]; rec["rot"] = np.asarray(leaves["rotation"]) & 3
    rec["x"] = leaves["x"]; rec["y"] = leaves["y"]; rec["val"] = np.asarray(leaves["active_tetromino"]) + 2
    rec["flags"] = np.where(np.asarray(leaves["game_over"], bool), N.F_GAME_OVER, 0)
    rec["holder_piece"] = N.NONE
    rec["bag"] = leaves["queue"]; rec["bag_index"] = leaves["queue_index"]; rec["score"] = leaves["score"]
    rng = np.zeros(B, N.RNG_REC)
    rng["key"] = leaves["rng_key"]
    board = np.asarray(leaves["board"]).reshape(B, config.height + p, config.width + 2 * p)
    e.set_state(board=board[:, : config.height, p: p + config.width].astype(np.uint8), rec=rec, rng=rng)
    slot.generation += 1


_FIELDS = ("rng_key", "board", "active_tetromino", "rotation", "x", "y", "queue", "queue_index", "game_over", "score")


def _state_from(leaves: dict, index, token) -> State:
    if index is None:
        return State(**{k: leaves[k] for k in _FIELDS}, _token=token)
    kw = {k: leaves[k][index] for k in _FIELDS}
    return State(**kw, _token=token)


def _leaves_of(state: State, batched: bool) -> dict:
    return {k: (np.asarray(getattr(state, k)) if batched else np.asarray(getattr(state, k))[None]) for k in _FIELDS}


def _ensure_resident(slot: _Slot, config, state: State, batched: bool):
    if state._token != (slot.id, slot.generation):
        _upload(slot, config, _leaves_of(state, batched))


def reset(tetrominoes, key, config, create_queue_fn=create_bag_queue, queue_fn=bag_queue_get_next_element):
    """→ (key, state, observation)   (reference envs/tetris_fn.py:318-367)"""
    _check(tetrominoes, config, create_queue_fn, queue_fn)
    key = np.asarray(key, np.uint32)
    slot = _slot(config, 1)
    slot.engine.reset(np.array([(int(key[0]) << 32) | int(key[1])], np.uint64))
    slot.generation += 1
    leaves = _download(slot, config)
    return split(key)[0], _state_from(leaves, 0, (slot.id, slot.generation)), slot.engine.observation()[0]


def step(tetrominoes, state: State, action, config, queue_fn=bag_queue_get_next_element):
    """→ (state, observation, reward, terminated, {"lines_cleared"})   (reference envs/tetris_fn.py:276-315)"""
    _check(tetrominoes, config, queue_fn)
    slot = _slot(config, 1)
    _ensure_resident(slot, config, state, False)
    rew, term, lines = slot.engine.step(np.array([int(action)], np.int32))
    slot.generation += 1
    leaves = _download(slot, config)
    return (_state_from(leaves, 0, (slot.id, slot.generation)), slot.engine.observation()[0], np.float32(rew[0]),
            np.bool_(term[0]), {"lines_cleared": np.int32(lines[0])})


def batched_reset(tetrominoes, keys, *, config, create_queue_fn=create_bag_queue,
                  queue_fn=bag_queue_get_next_element, batch_size: int = 1):
    """→ (keys[B,2], states, observations[B,H,W])   (reference envs/tetris_fn.py:438-467; like the reference,
    `batch_size` is accepted and unused — the batch is keys.shape[0])."""
    _check(tetrominoes, config, create_queue_fn, queue_fn)
    keys = np.asarray(keys, np.uint32).reshape(-1, 2)
    slot = _slot(config, keys.shape[0])
    slot.engine.reset((keys[:, 0].astype(np.uint64) << np.uint64(32)) | keys[:, 1].astype(np.uint64))
    slot.generation += 1
    leaves = _download(slot, config)
    new_keys = np.array([split(k)[0] for k in keys], np.uint32)
    return new_keys, _state_from(leaves, None, (slot.id, slot.generation)), slot.engine.observation()


def batched_step(tetrominoes, states: State, actions, *, config, queue_fn=bag_queue_get_next_element):
    """→ (states, observations, rewards f32[B], terminated bool[B], {"lines_cleared": i32[B]})
    (reference envs/tetris_fn.py:416-435)"""
    _check(tetrominoes, config, queue_fn)
    actions = np.asarray(actions, np.int32).reshape(-1)
    slot = _slot(config, actions.shape[0])
    _ensure_resident(slot, config, states, True)
    rew, term, lines = slot.engine.step(actions)
    slot.generation += 1
    leaves = _download(slot, config)
    return (_state_from(leaves, None, (slot.id, slot.generation)), slot.engine.observation(), rew, term.astype(bool),
            {"lines_cleared": lines.astype(np.int32)})


def get_observation(board, x, y, active_tetromino, rotation, game_over, tetrominoes, config):
    """Host restatement of reference envs/tetris_fn.py:137-158 for a single state (tests / tooling)."""
    b = (np.asarray(board) > 0).astype(np.int8)
    if not bool(game_over):
        m = tetrominoes.matrices[int(active_tetromino), int(rotation)]
        b = b.copy()
        b[int(y): int(y) + 4, int(x): int(x) + 4] -= m[: b.shape[0] - int(y), : b.shape[1] - int(x)]
    p = config.padding
    return b[0:-p, p:-p]
