"""Functional `reset / step / batched_reset / batched_step` with the reference's signatures
(tetris_gymnasium/envs/tetris_fn.py:276-467), evaluated by the HIP engine (rule-set FUNCTIONAL).

Pure-functional contract kept: inputs are never mutated, a new `State` is returned.  A returned State
carries a private token naming the engine generation that produced it; passing it straight back (the
usual rollout loop) skips the host→device upload.  Stock TETROMINOES and the stock 7-bag queue only —
anything else is rejected explicitly.

`EnvConfig.padding` (reference functional/core.py:23,59,79,198): any value >= 3.  The engine keeps boards as playfields (the
bedrock frame is implicit) and piece positions in the coordinates of a 4-wide frame; a State of another padding is that
state with x moved by `padding - 4` and a wider or narrower frame of 1-cells around the board — exact as long as every
4x4 window the rules read lies inside the framed array, which holds for every position play can reach once padding >= 3
(a hand-made position whose window would be clamped by lax.dynamic_slice under one of the two paddings is refused).  Below
3 the reference's own windows are clamped in ordinary play (an upright I against the left wall): refused.

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

from ..functional.prng import PRNGKey, permutation7, split, split_batch, threefry2x32, threefry2x32_batch  # noqa: E402,F401


# ---------------------------------------------------------------------------------------------------
# Device-resident values of the batched functions.  Like the reference's jax arrays, what batched_reset / batched_step
# return lives on the device until somebody looks at it: every returned value owns a device copy (pooled buffers,
# device-to-device on the engine's stream — no synchronisation, no PCIe traffic in a rollout loop) and turns into NumPy
# on first use (`np.asarray`, indexing, arithmetic, attribute access).  A State handed back to batched_step is restored
# device-to-device unless it is the one the engine already holds (`_token`).
class _Source:
    """One download serving several LazyArray leaves (a State's three buffers, the three fields of tg_step_out)."""

    def __init__(self, fetch):
        self._fetch, self._vals = fetch, None
        self.downloads = 0

    def get(self, key):
        if self._vals is None:
            self._vals = self._fetch()
            self._fetch = None  # (drops the device buffers the closure holds)
            self.downloads += 1
        return self._vals[key]


class LazyArray:
    """An array that is still on the device.  Behaves like the NumPy array it becomes on first use."""

    def __init__(self, source: _Source, key, shape, dtype, device_view=None):
        self._source, self._key, self._shape, self._dtype, self._view = source, key, tuple(shape), np.dtype(dtype), device_view

    # cheap metadata: no download
    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._dtype)
    ndim = property(lambda self: len(self._shape))
    size = property(lambda self: int(np.prod(self._shape, dtype=np.int64)))

    def __len__(self):
        return self._shape[0]

    @property
    def materialized(self) -> bool:
        return self._source._vals is not None

    def device_view(self):
        """The DeviceArray behind this value (zero-copy hand-over to torch / cupy), or None once it only exists on the host."""
        return None if self.materialized else self._view

    def numpy(self) -> np.ndarray:
        return self._source.get(self._key)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, i):
        return self.numpy()[i]

    def __iter__(self):
        return iter(self.numpy())

    def __getattr__(self, name):  # everything else an ndarray offers (sum, any, astype, reshape, copy, ...)
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    def __repr__(self):
        return repr(self.numpy()) if self.materialized else f"LazyArray(on device, shape={self._shape}, dtype={self._dtype})"

    def __bool__(self):
        return bool(self.numpy())

    def __int__(self):
        return int(self.numpy())

    def __float__(self):
        return float(self.numpy())

    def __index__(self):
        return int(self.numpy())


def _binop(name):
    def f(self, other):
        return getattr(self.numpy(), name)(np.asarray(other) if isinstance(other, LazyArray) else other)
    return f


for _n in ("add", "sub", "mul", "truediv", "floordiv", "mod", "and", "or", "xor", "lt", "le", "gt", "ge", "eq", "ne",
           "radd", "rsub", "rmul", "rand", "ror", "rxor", "neg", "invert", "abs"):
    if _n in ("neg", "invert", "abs"):
        setattr(LazyArray, f"__{_n}__", (lambda nm: lambda self: getattr(self.numpy(), nm)())(f"__{_n}__"))
    else:
        setattr(LazyArray, f"__{_n}__", _binop(f"__{_n}__"))
LazyArray.__hash__ = None

_engines: dict = {}
_ids = itertools.count(1)
_ENGINE_PADDING = 4  # the frame the engine's piece coordinates are in (include/tetris_gym_hip.h: tg_config.padding)


class _Slot:
    def __init__(self, config: EnvConfig, batch: int):
        self.engine = Engine(batch, config.width, config.height, ruleset=N.RULESET_FUNCTIONAL,
                             gravity=config.gravity_enabled, queue_size=int(config.queue_size), obs_mode=N.OBS_FN,
                             padding=_ENGINE_PADDING)
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
    if int(config.padding) < 3 or not 1 <= int(config.queue_size) <= 7:
        raise NotImplementedError("the HIP engine needs padding >= 3 (below that lax.dynamic_slice clamps the reference's 4x4 "
                                  "windows in ordinary play) and a bag length queue_size in 1..7 (stock tetromino set)")


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
        rotation=rec["rot"].astype(np.int32), x=rec["x"].astype(np.int32) + (int(config.padding) - _ENGINE_PADDING),
        y=rec["y"].astype(np.int32),
        queue=rec["bag"][:, : config.queue_size].astype(np.int32), queue_index=rec["bag_index"].astype(np.int32),
        game_over=(rec["flags"] & N.F_GAME_OVER).astype(bool), score=rec["score"].astype(np.float32))


def _upload(slot: _Slot, config, leaves: dict):
    e = slot.engine
    B, p = e.B, int(config.padding)
    rec = np.zeros(B, N.ENV_REC)
    rec["piece"] = leaves["active_tetromino"]; rec["rot"] = np.asarray(leaves["rotation"]) & 3
    x, y = np.asarray(leaves["x"], np.int64) - (p - _ENGINE_PADDING), np.asarray(leaves["y"], np.int64)
    if p != _ENGINE_PADDING:  # (x is now in the engine's frame)  a window that one of the two frames would clamp: not the same state
        lo, hi, ymax = max(0, _ENGINE_PADDING - p), config.width + min(_ENGINE_PADDING, p), config.height + min(0, p - _ENGINE_PADDING)
        if ((x < lo) | (x > hi) | (y < 0) | (y > ymax)).any():
            raise NotImplementedError(f"padding={p}: a piece whose 4x4 window leaves the framed board (x outside {lo + p - 4}..{hi + p - 4} "
                                      f"or y outside 0..{ymax}) is clamped differently by the reference and by the engine's 4-wide frame")
    rec["x"] = x; rec["y"] = y; rec["val"] = np.asarray(leaves["active_tetromino"]) + 2
    rec["flags"] = np.where(np.asarray(leaves["game_over"], bool), N.F_GAME_OVER, 0)
    rec["holder_piece"] = N.NONE
    rec["bag"][:, : config.queue_size] = np.asarray(leaves["queue"]).reshape(B, -1)
    rec["bag_index"] = leaves["queue_index"]; rec["score"] = leaves["score"]
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


def _leaves_from_buffers(config, B, board, rec, rng) -> dict:
    return dict(
        rng_key=rng["key"].copy(), board=_pad(board, config), active_tetromino=rec["piece"].astype(np.int32),
        rotation=rec["rot"].astype(np.int32), x=rec["x"].astype(np.int32) + (int(config.padding) - _ENGINE_PADDING),
        y=rec["y"].astype(np.int32),
        queue=rec["bag"][:, : config.queue_size].astype(np.int32), queue_index=rec["bag_index"].astype(np.int32),
        game_over=(rec["flags"] & N.F_GAME_OVER).astype(bool), score=rec["score"].astype(np.float32))


_LEAF_SPEC = {  # leaf → (trailing shape as a function of the config, dtype)
    "rng_key": (lambda c: (2,), np.uint32), "board": (lambda c: (c.height + c.padding, c.width + 2 * c.padding), np.int8),
    "active_tetromino": (lambda c: (), np.int32), "rotation": (lambda c: (), np.int32), "x": (lambda c: (), np.int32),
    "y": (lambda c: (), np.int32), "queue": (lambda c: (int(c.queue_size),), np.int32), "queue_index": (lambda c: (), np.int32),
    "game_over": (lambda c: (), np.bool_), "score": (lambda c: (), np.float32)}


def _device_state(slot: _Slot, config) -> State:
    """The engine's current state as a State whose leaves are still on the device (own device copy: stays valid — and can be
    handed back to batched_step — however far the engine moves on)."""
    e = slot.engine
    bufs = e.get_state_device()
    B = e.B

    def fetch(bufs=bufs):
        board = e.dev_download(bufs[0].ptr, np.empty((B, e.H, e.W), np.uint8))
        rec = e.dev_download(bufs[1].ptr, np.empty(B, N.ENV_REC))
        rng = e.dev_download(bufs[2].ptr, np.empty(B, N.RNG_REC))
        return _leaves_from_buffers(config, B, board, rec, rng)

    src = _Source(fetch)
    st = State(**{k: LazyArray(src, k, (B,) + shp(config), dt) for k, (shp, dt) in _LEAF_SPEC.items()},
               _token=(slot.id, slot.generation))
    st._device = (slot.id, bufs, src)  # device copy for a device-to-device restore; dropped with the State
    return st


def _device_outputs(slot: _Slot, config):
    """(observation, reward, terminated, lines) of the last transition as device-resident values."""
    from ..engine import DeviceArray, DeviceBuffer

    e = slot.engine
    B = e.B
    src_o, n_o = e.device_ptr(N.BUF_OBS)
    src_r, n_r = e.device_ptr(N.BUF_STEP_OUT)
    bo, br = DeviceBuffer(e, n_o), DeviceBuffer(e, n_r)
    e.dev_copy(bo.ptr, src_o, n_o)
    e.dev_copy(br.ptr, src_r, n_r)
    so = _Source(lambda bo=bo: {"obs": e.dev_download(bo.ptr, np.empty((B, config.height, config.width), np.int8))})

    def fetch_out(br=br):
        out = e.dev_download(br.ptr, np.empty(B, N.STEP_OUT))
        return {"reward": np.ascontiguousarray(out["reward"]), "terminated": out["terminated"].astype(bool),
                "lines": out["lines"].astype(np.int32)}

    sr = _Source(fetch_out)
    st = N.STEP_OUT.itemsize
    obs = LazyArray(so, "obs", (B, config.height, config.width), np.int8,
                    DeviceArray(e, bo.ptr, (B, config.height, config.width), np.int8, owner=bo))
    rew = LazyArray(sr, "reward", (B,), np.float32, DeviceArray(e, br.ptr, (B,), np.float32, strides=(st,), owner=br))
    term = LazyArray(sr, "terminated", (B,), np.bool_, DeviceArray(e, br.ptr + 4, (B,), np.uint8, strides=(st,), owner=br))
    lines = LazyArray(sr, "lines", (B,), np.int32, DeviceArray(e, br.ptr + 5, (B,), np.uint8, strides=(st,), owner=br))
    return obs, rew, term, lines


def _ensure_resident_batched(slot: _Slot, config, states: State):
    if states._token == (slot.id, slot.generation):
        return
    dev = getattr(states, "_device", None)
    if dev is not None and dev[0] == slot.id:  # an earlier State of this engine (leaves are immutable): device → device
        slot.engine.set_state_device(*dev[1])
        slot.generation += 1
        return
    _upload(slot, config, _leaves_of(states, True))


def batched_reset(tetrominoes, keys, *, config, create_queue_fn=create_bag_queue,
                  queue_fn=bag_queue_get_next_element, batch_size: int = 1):
    """→ (keys[B,2], states, observations[B,H,W])   (reference envs/tetris_fn.py:438-467; like the reference,
    `batch_size` is accepted and unused — the batch is keys.shape[0]).  States and observations stay on the device
    until read (LazyArray)."""
    _check(tetrominoes, config, create_queue_fn, queue_fn)
    keys = np.asarray(keys, np.uint32).reshape(-1, 2)
    slot = _slot(config, keys.shape[0])
    slot.engine.reset((keys[:, 0].astype(np.uint64) << np.uint64(32)) | keys[:, 1].astype(np.uint64))
    slot.generation += 1
    obs = _device_outputs(slot, config)[0]
    return split_batch(keys, 0), _device_state(slot, config), obs


def batched_step(tetrominoes, states: State, actions, *, config, queue_fn=bag_queue_get_next_element):
    """→ (states, observations, rewards f32[B], terminated bool[B], {"lines_cleared": i32[B]})
    (reference envs/tetris_fn.py:416-435).  Nothing is downloaded and nothing waits for the GPU: every returned value is
    a LazyArray that owns a device copy and becomes NumPy when it is first read."""
    _check(tetrominoes, config, queue_fn)
    actions = np.asarray(actions, np.int32).reshape(-1)
    slot = _slot(config, actions.shape[0])
    _ensure_resident_batched(slot, config, states)
    slot.engine.step_async(actions)
    slot.generation += 1
    obs, rew, term, lines = _device_outputs(slot, config)
    return _device_state(slot, config), obs, rew, term, {"lines_cleared": lines}


def get_observation(board, x, y, active_tetromino, rotation, game_over, tetrominoes, config):
    """Host restatement of reference envs/tetris_fn.py:137-158 for a single state (tests / tooling)."""
    b = (np.asarray(board) > 0).astype(np.int8)
    if not bool(game_over):
        m = tetrominoes.matrices[int(active_tetromino), int(rotation)]
        b = b.copy()
        b[int(y): int(y) + 4, int(x): int(x) + 4] -= m[: b.shape[0] - int(y), : b.shape[1] - int(x)]
    p = config.padding
    return b[0:-p, p:-p]


# ---------------------------------------------------------------------------------------------------------------------
# The two remaining functions of reference envs/tetris_fn.py as host-side NumPy for ONE state (arguments and returns as in
# the reference; `queue_fn` may be any callable with the reference's signature — this is plain host code).  The engine runs
# the same rules inside its step kernel; these exist for callers and tests that use the pieces on their own.
def place_active_tetromino(config, tetrominoes, state: State, queue_fn=bag_queue_get_next_element):
    """→ (state, reward, lines_cleared): lock the active piece, clear rows, spawn the next piece of the queue, detect game
    over, advance the key (reference envs/tetris_fn.py:370-413)."""
    from ..functional import core as fc

    board, reward, lines = fc.lock_active_tetromino(config, tetrominoes, state.board, state.active_tetromino, state.rotation,
                                                    state.x, state.y)
    piece, queue, queue_index, _ = queue_fn(config, state.queue, state.queue_index, state.rng_key)
    x, y = fc.get_initial_x_y(config, tetrominoes, piece)
    over = fc.check_game_over(tetrominoes, board, piece, 0, x, y)
    new = State(rng_key=split(np.asarray(state.rng_key, np.uint32))[0], board=board, active_tetromino=np.int32(piece),
                rotation=np.int32(0), x=np.int32(x), y=np.int32(y), queue=np.asarray(queue, np.int32),
                queue_index=np.int32(queue_index), game_over=np.bool_(over), score=state.score)
    return new, np.int32(reward), np.int32(lines)


def update_state(action, config, state: State, queue_fn=bag_queue_get_next_element, tetrominoes=TETROMINOES):
    """→ (state, lines_cleared): one action applied to a state that is not game over (reference envs/tetris_fn.py:161-270):
    move / rotate if the target is free (ids outside 0..6 are clamped the way lax.switch clamps its index), soft drop scores
    a row, hard drop two per row and always locks, gravity pulls one row and locks the piece when it cannot fall."""
    from ..functional import core as fc

    a = min(max(int(action), 0), 6)
    x, y, rot = int(state.x), int(state.y), int(state.rotation)
    board, piece = state.board, int(state.active_tetromino)
    mat = tetrominoes.matrices[piece, rot & 3]
    drop_reward = 0
    if a == 0 and not fc.collision(board, mat, x - 1, y):
        x -= 1
    elif a == 1 and not fc.collision(board, mat, x + 1, y):
        x += 1
    elif a == 2 and not fc.collision(board, mat, x, y + 1):
        y, drop_reward = y + 1, 1
    elif a in (3, 4):
        nrot = (rot + (1 if a == 4 else -1)) % 4
        if not fc.collision(board, tetrominoes.matrices[piece, nrot], x, y):
            rot, mat = nrot, tetrominoes.matrices[piece, nrot]
    elif a == 6:
        y, drop_reward = fc.hard_drop(board, mat, x, y)
    y_gravity = fc.graviy_step(tetrominoes, board, piece, rot, x, y) if config.gravity_enabled else y
    should_lock = (y_gravity == y) and bool(config.gravity_enabled)
    mid = State(rng_key=state.rng_key, board=board, active_tetromino=state.active_tetromino, rotation=np.int32(rot),
                x=np.int32(x), y=np.int32(y_gravity), queue=state.queue, queue_index=state.queue_index,
                game_over=np.bool_(False), score=state.score)
    new, lock_reward, lines = (place_active_tetromino(config, tetrominoes, mid, queue_fn)
                               if (should_lock or int(action) == 6) else (mid, np.int32(0), np.int32(0)))
    new = new.replace(score=np.float32(np.float32(new.score) + np.float32(int(drop_reward)) + np.float32(int(lock_reward))))
    return new, lines
