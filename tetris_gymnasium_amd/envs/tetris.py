"""`Tetris`: the reference's Gymnasium environment surface (tetris_gymnasium/envs/tetris.py:36-708) on top of
the HIP engine (rule-set NUMPY, one env).  Same constructor kwargs, `reset`/`step` signatures, observation
dict, rewards, and the attributes the reference's wrappers and tests reach into (`board, x, y,
active_tetromino, queue, holder, randomizer, has_swapped, game_over, padding, width_padded, ...`).

The state lives on the device.  The attributes are host-side views of it that are filled in LAZILY: the first
read after a transition downloads the state once (`tg_get_state`), a write (or a read of a mutable object —
`env.board[...] = ...`, `env.queue.queue`, `env.active_tetromino`) marks the host copy as possibly changed and
the next `step()` uploads it first (`tg_set_state`), so poking `env.x`, `env.board` or `env.active_tetromino`
between steps works like in the reference's tests.  A plain `reset(); while ...: step(a)` loop touches none of
them: a step is then ONE call into the C ABI (tg_step_fetch_many: action upload, kernel launch, the copies of
TG_BUF_OBS and TG_BUF_STEP_OUT, one wait) — round 3 made six synchronous round trips per step.  This is the
drop-in / debugging surface; throughput comes from `TetrisVectorEnv` / `Engine`.
"""
from __future__ import annotations

import copy
from collections import deque
from dataclasses import dataclass, fields
from typing import Any, List

import numpy as np

from .. import _native as N
from .._gym_compat import EnvBase, spaces
from ..components.tetromino import COLORS, Pixel, Tetromino, identify, make_tetromino
from ..components.tetromino_holder import TetrominoHolder
from ..components.tetromino_queue import TetrominoQueue
from ..components.tetromino_randomizer import BagRandomizer
from ..engine import Engine
from ..mappings.actions import ActionsMapping
from ..mappings.rewards import RewardsMapping


class _QueueView(TetrominoQueue):
    """`Tetris.queue`: a TetrominoQueue (reference components/tetromino_queue.py) whose content mirrors the device-side queue
    of the env — the randomizer behind it runs on the device."""

    def __init__(self, size):
        super().__init__(None, size)


class _HolderView(TetrominoHolder):
    """`Tetris.holder`: a TetrominoHolder (reference components/tetromino_holder.py) mirroring the device-side hold slot."""


class _RandomizerView(BagRandomizer):
    """`Tetris.randomizer`: a BagRandomizer (reference components/tetromino_randomizer.py:49-102) whose bag / index mirror the
    device-side generator of the env; `pcg64_state` is that generator as a tg_rng_rec NumPy record."""

    def __init__(self, size=7):
        super().__init__(size)
        self.pcg64_state = None


@dataclass
class TetrisState:
    board: np.ndarray
    rec: np.ndarray
    rng: np.ndarray


def _state_attr(name, mutable):
    """a Tetris attribute backed by the lazily downloaded device state (see the module docstring)"""

    def get(self):
        self._sync_host()
        if mutable and not self._dirty and self._handed is None:
            # the caller may change the object in place: remember what the mutable objects look like now — the next step() uploads
            # the host copy only if they differ then (a wrapper that merely LOOKS at env.board / env.active_tetromino every step, as
            # the reference's grouped and feature wrappers do, keeps the one-call step path)
            self._handed = self._fingerprint()
        return self._hs[name]

    def set(self, value):
        self._sync_host()
        self._hs[name] = value
        self._dirty = True

    return property(get, set)


class Tetris(EnvBase):
    metadata = {"render_modes": ["rgb_array", "ansi"], "render_fps": 1}
    board = _state_attr("board", True)
    x = _state_attr("x", False)
    y = _state_attr("y", False)
    active_tetromino = _state_attr("active_tetromino", True)
    has_swapped = _state_attr("has_swapped", False)
    game_over = _state_attr("game_over", False)
    queue = _state_attr("queue", True)
    holder = _state_attr("holder", True)
    randomizer = _state_attr("randomizer", True)

    BASE_PIXELS = [Pixel(0, [0, 0, 0]), Pixel(1, [128, 128, 128])]
    TETROMINOES = [Tetromino(i, list(COLORS[i]), make_tetromino(i, 0, 1).matrix) for i in range(7)]

    def __init__(self, render_mode=None, width=10, height=20, gravity=True, actions_mapping=ActionsMapping(),
                 rewards_mapping=RewardsMapping(), queue=None, holder=None, randomizer=None, base_pixels=None,
                 tetrominoes=None, render_upscale: int = 10, queue_size: int = 4, device: int = 0):
        if any(a is not None for a in (queue, holder, randomizer, base_pixels, tetrominoes)):
            raise NotImplementedError("custom queue/holder/randomizer/base_pixels/tetrominoes are not supported by the "
                                      "HIP engine (they are also broken in the reference: envs/tetris.py:114-144)")
        if len(fields(actions_mapping)) != 8 or len(fields(rewards_mapping)) != 4:
            raise NotImplementedError("mappings extended with further fields have no meaning to the HIP engine (the reference's "
                                      "step reads the eight actions / four rewards of mappings/*.py)")
        self._hs, self._stale, self._dirty, self._engine, self._handed = {}, False, False, None, None
        self._deferred = None  # a wrapper's pending write-back of this env's state (_defer / _settle)
        self._version = 0  # counts the changes of the DEVICE state (step / reset / set_state / an upload of host-side edits): wrappers/_device.py
        self.game_over = False
        self.height, self.width = int(height), int(width)
        self.base_pixels = copy.deepcopy(self.BASE_PIXELS)
        self.tetrominoes: List[Tetromino] = [make_tetromino(i) for i in range(7)]  # ids/values offset by 2
        self.pixels = self.base_pixels + [Pixel(t.id, t.color_rgb) for t in self.tetrominoes]
        self.padding = 4
        self.width_padded = self.width + 2 * self.padding
        self.height_padded = self.height + self.padding
        self.gravity_enabled = bool(gravity)
        self.actions, self.rewards = actions_mapping, rewards_mapping
        self.queue = _QueueView(queue_size)
        self.holder = _HolderView(1)
        self.randomizer = _RandomizerView(7)
        self.has_swapped = False
        self.x = self.y = 0
        self.active_tetromino: Tetromino = None
        self.board = self.create_board()
        # (the two mappings are per-handle data of the step kernel: envs/tetris.py:83-84, 189-190)
        self._engine = Engine(1, self.width, self.height, ruleset=N.RULESET_NUMPY, gravity=gravity,
                              queue_size=queue_size, obs_mode=N.OBS_DICT, device=device,
                              actions_mapping=actions_mapping, rewards_mapping=rewards_mapping)
        self._rec = None
        self._init_fast_step()
        self._stale, self._dirty = True, False  # (whatever the placeholders above were: the device state is the truth)
        hi = len(self.pixels)
        self.observation_space = spaces.Dict({
            "board": spaces.Box(low=0, high=hi, shape=(self.height_padded, self.width_padded), dtype=np.uint8),
            "active_tetromino_mask": spaces.Box(low=0, high=1, shape=(self.height_padded, self.width_padded),
                                                dtype=np.uint8),
            "holder": spaces.Box(low=0, high=hi, shape=(self.padding, self.padding * self.holder.size), dtype=np.uint8),
            "queue": spaces.Box(low=0, high=hi, shape=(self.padding, self.padding * self.queue.size), dtype=np.uint8),
        })
        self.action_space = spaces.Discrete(len(fields(self.actions)))
        self.reward_range = (min(vars(self.rewards).values()), max(vars(self.rewards).values()))
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self.render_mode = render_mode
        self.render_scaling_factor = render_upscale

    # ---- host <-> device state ----------------------------------------------------------------
    def _defer(self, writeback):
        """A wrapper advanced its mirror of this env (wrappers/grouped.py steps an auxiliary engine): the env's state is now what
        `writeback()` will upload.  It runs before anything reads or changes this env (_settle) — in a loop that only talks to the
        wrapper it never runs, and the state is not copied back and forth once per step."""
        self._deferred = writeback
        self._stale, self._dirty, self._handed = True, False, None
        self._version += 1

    def _settle(self):
        d, self._deferred = self._deferred, None
        if d is not None:
            d()

    def _sync_host(self):
        """fill the host-side attributes from the device state if a transition has run since the last download"""
        self._settle()
        if self._stale:
            self._stale = False
            self._pull()

    def _fingerprint(self):
        """what the mutable host-side objects hold (board bytes, active piece, queue, holder, bag): cheap to compare"""
        hs = self._hs
        t = hs["active_tetromino"]
        held = hs["holder"].get_tetrominoes()
        return (hs["board"].tobytes(), None if t is None else (t.id, t.matrix.tobytes()), tuple(int(v) for v in hs["queue"].queue),
                tuple((h.id, h.matrix.tobytes()) for h in held), np.asarray(hs["randomizer"].bag).tobytes(), int(hs["randomizer"].index))

    def _changed_in_place(self) -> bool:
        """a mutable attribute was handed out since the last download: did the caller change it?"""
        if self._handed is None:
            return False
        old, self._handed = self._handed, None
        changed = (not self._stale) and old != self._fingerprint()
        if not changed and not self._stale:
            # (nothing changed and the host copy is still current: the caller may still hold the reference it was given —
            # `b = env.board; env.render(); b[...] = v; env.step()` edits the live array in the reference — so the
            # fingerprint stays armed; it is dropped only when the host copy is replaced: _pull, _push, a step)
            self._handed = old
        return changed

    def _flush(self):
        """host-side edits (attribute writes, in-place changes of handed-out objects) → device"""
        self._settle()
        if self._dirty or self._changed_in_place():
            self._push()

    def _ro(self, name):
        """read an attribute without marking the host copy as changed (the helpers below only look)"""
        self._sync_host()
        return self._hs[name]

    def _push(self):
        """upload the host-side attributes (they are current: every write path went through _sync_host first)"""
        e, hs = self._engine, self._hs
        rec, rng = self._rec, hs["randomizer"].pcg64_state
        idx, rot, val = identify(hs["active_tetromino"])
        rec["piece"], rec["rot"], rec["val"] = idx, rot, val
        rec["x"], rec["y"] = hs["x"], hs["y"]
        rec["flags"] = (N.F_GAME_OVER if hs["game_over"] else 0) | (N.F_HAS_SWAPPED if hs["has_swapped"] else 0)
        held = hs["holder"].get_tetrominoes()
        if held:
            hi_, hr, hv = identify(held[0])
            rec["holder_piece"], rec["holder_rot"], rec["holder_val"] = hi_, hr, hv
        else:
            rec["holder_piece"] = N.NONE
        q = list(hs["queue"].queue)
        rec["queue"][0, : len(q)] = q
        rec["bag"][0] = hs["randomizer"].bag
        rec["bag_index"] = hs["randomizer"].index
        p = self.padding
        board = np.ascontiguousarray(hs["board"][: self.height, p: p + self.width], np.uint8)[None]
        self._version += 1
        e.set_state(board=board, rec=rec, rng=rng)  # (tg_set_state re-packs TG_BUF_OBS itself)
        self._dirty, self._handed = False, None

    def _pull(self):
        board, rec, rng = self._engine.get_state()
        r, hs = rec[0], self._hs
        self._rec = rec
        hs["board"] = self.create_board()
        p = self.padding
        hs["board"][: self.height, p: p + self.width] = board[0]
        hs["active_tetromino"] = make_tetromino(int(r["piece"]), int(r["rot"]), int(r["val"]))
        hs["x"], hs["y"] = int(r["x"]), int(r["y"])
        hs["game_over"] = bool(r["flags"] & N.F_GAME_OVER)
        hs["has_swapped"] = bool(r["flags"] & N.F_HAS_SWAPPED)
        hs["queue"].queue = deque((np.int8(v) for v in r["queue"][: hs["queue"].size]), maxlen=hs["queue"].size)
        hs["holder"].queue = deque(maxlen=1)
        if r["holder_piece"] != N.NONE:
            hs["holder"].queue.append(make_tetromino(int(r["holder_piece"]), int(r["holder_rot"]), int(r["holder_val"])))
        hs["randomizer"].bag = r["bag"].astype(np.int8)
        hs["randomizer"].index = int(r["bag_index"])
        hs["randomizer"].pcg64_state = rng.copy()

    def _obs(self) -> "dict[str, Any]":
        # TG_BUF_OBS is what the step kernel wrote (after a transition) or what tg_reset / tg_set_state re-packed
        self._settle()
        o = self._engine.observation()
        return {k: v[0].copy() for k, v in o.items()}

    def _init_fast_step(self):
        """ctypes arguments of the one-call step: [TG_BUF_OBS, TG_BUF_STEP_OUT] of the one-env engine into two host arrays"""
        import ctypes as C

        e = self._engine
        cells, pp = self.height_padded * self.width_padded, self.padding * self.padding
        # page-locked: a copy to or from pageable memory is staged and waited for inside the runtime (measured: 46 us per step
        # against 25 with these three buffers page-locked; the page-locked action is read in place by the kernel; tools/single_env_latency.py)
        # (two copies: joining the observation and the 8-byte record into ONE copy across the 16 KiB between them on the device was
        # measured 3.7 us SLOWER than two small ones)
        self._obs_flat = e.pinned_bytes(2 * cells + pp + pp * e.Q)
        self._out1 = e.pinned_bytes(N.STEP_OUT.itemsize).view(N.STEP_OUT)
        self._act1 = e.pinned_bytes(4).view(np.int32)
        self._act1[0] = 0
        self._fs = ((C.c_void_p * 1)(e.h.value), (C.c_void_p * 1)(self._act1.ctypes.data), (C.c_int * 2)(N.BUF_OBS, N.BUF_STEP_OUT),
                    (C.c_void_p * 2)(self._obs_flat.ctypes.data, self._out1.ctypes.data),
                    (C.c_size_t * 2)(self._obs_flat.nbytes, self._out1.nbytes))
        hp, wp, p = self.height_padded, self.width_padded, self.padding
        f = self._obs_flat
        self._obs_views = {"board": f[:cells].reshape(hp, wp), "active_tetromino_mask": f[cells:2 * cells].reshape(hp, wp),
                           "holder": f[2 * cells:2 * cells + pp].reshape(p, p), "queue": f[2 * cells + pp:].reshape(p, p * e.Q)}

    # ---- gymnasium API --------------------------------------------------------------------------
    def step(self, action):
        """reference envs/tetris.py:203-272"""
        assert self.action_space.contains(action), f"{action!r} ({type(action)}) invalid"
        self._flush()
        e = self._engine
        hs, acts, which, dst, nbytes = self._fs
        self._act1[0] = int(action)
        self._version += 1
        e._chk(e.lib.tg_step_fetch_many(1, hs, acts, N.MEM_HOST, 2, which, dst, nbytes, None), "tg_step_fetch_many")
        self._stale, self._handed = True, None
        term, lines, fl = bool(self._out1["terminated"][0]), int(self._out1["lines"][0]), int(self._out1["flags"][0])
        # the reward as the reference's Python computes it (envs/tetris.py:461-474): int score + the mapping's own numbers
        reward = 0
        if fl & N.OUT_COMMITTED:
            reward = self.rewards.game_over if term else self.score(lines) + self.rewards.alife
        return {k: v.copy() for k, v in self._obs_views.items()}, reward, term, False, {"lines_cleared": lines}

    def reset(self, *, seed: "int | None" = None, options: "dict[str, Any] | None" = None):
        """reference envs/tetris.py:274-307 (seed=None or 0 keeps the running generator)"""
        try:
            super().reset(seed=seed, options=options)
        except TypeError:
            pass
        self._settle()  # (an unseeded reset continues the generator of the state as it stands)
        self._version += 1
        self._engine.reset(None if not seed else np.array([int(seed)], np.uint64))
        self._stale, self._dirty, self._handed = True, False, None
        return self._obs(), {"lines_cleared": 0}

    # ---- helpers the reference's wrappers / tests call ------------------------------------------
    def create_board(self) -> np.ndarray:
        """reference envs/tetris.py:623-641: the empty playfield inside its bedrock frame (a copy of a template: np.pad costs 25 us,
        and the host copy of the board is rebuilt after every transition somebody looks at)"""
        t = self.__dict__.get("_board_tmpl")
        if t is None or t.shape != (self.height + self.padding, self.width + 2 * self.padding):
            t = np.ones((self.height + self.padding, self.width + 2 * self.padding), np.uint8)
            t[: self.height, self.padding: self.padding + self.width] = 0
            self.__dict__["_board_tmpl"] = t
        return t.copy()

    def get_tetromino_slices(self, tetromino, x, y):
        h, w = tetromino.matrix.shape
        return (slice(y, y + h), slice(x, x + w))

    def collision(self, tetromino, x, y) -> bool:
        sub = self._ro("board")[self.get_tetromino_slices(tetromino, x, y)]
        return bool(np.any(sub[tetromino.matrix > 0] > 0))

    def rotate(self, tetromino, clockwise=True):
        return Tetromino(tetromino.id, tetromino.color_rgb, np.rot90(tetromino.matrix, k=(1 if clockwise else -1)))

    def crop_padding(self, matrix):
        return matrix[0: -self.padding, self.padding: -self.padding]

    def project_tetromino(self, tetromino=None, x=None, y=None):
        tetromino = self._ro("active_tetromino") if tetromino is None else tetromino
        x = self._ro("x") if x is None else x
        y = self._ro("y") if y is None else y
        proj = self._ro("board").copy()
        if self.collision(tetromino, x, y):
            return proj
        proj[self.get_tetromino_slices(tetromino, x, y)] += tetromino.matrix
        return proj

    def score(self, rows_cleared) -> int:
        return (rows_cleared ** 2) * self.width

    # (further helpers of the reference class that foreign wrappers call on HOST arrays: wrappers/grouped.py:174 clears the rows of a
    # projected copy, wrappers/observation.py:82 renders `_get_obs()`; the env's own transition — incl. its row clears — is the step kernel)
    def clear_filled_rows(self, board):
        """reference envs/tetris.py:481-512: rows without a free cell that are not bedrock leave, empty rows enter at the top (in place)"""
        full = np.all(board != 0, axis=1) & np.any(board != 1, axis=1)
        n = int(np.count_nonzero(full))
        if n:
            board[n:] = board[~full]
            board[:n] = 1
            board[:n, self.padding: self.padding + self.width] = 0
        return board, n

    def _get_obs(self):
        """reference envs/tetris.py:566-615 (packed on the device; host-side edits of the attributes are uploaded first)"""
        self._flush()
        return self._obs()

    def _get_info(self):
        return {"lines_cleared": 0}

    def get_rgb(self, observation):
        """reference envs/tetris.py:309-343: an observation dict as ONE image — board left, queue strip top right, holder bottom right"""
        b, h, q = (np.asarray(observation[k]) for k in ("board", "holder", "queue"))
        side = np.ones((b.shape[0], max(h.shape[1], q.shape[1])), np.int64)
        side[: q.shape[0], : q.shape[1]] = q
        side[b.shape[0] - h.shape[0]:, : h.shape[1]] = h
        return np.array([p.color_rgb for p in self.pixels], np.uint8)[np.hstack((b.astype(np.int64), side))]

    def get_state(self) -> TetrisState:
        """Clone the full state incl. the PCG64 generator (reference envs/tetris.py:694-708)."""
        self._flush()
        b, r, g = self._engine.get_state()
        return TetrisState(b.copy(), r.copy(), g.copy())

    def set_state(self, state: TetrisState) -> None:
        self._deferred = None  # (everything a pending write-back would bring is replaced)
        self._version += 1
        self._engine.set_state(board=state.board, rec=state.rec, rng=state.rng)
        self._stale, self._dirty, self._handed = True, False, None

    def render(self):
        if self.render_mode == "ansi":
            proj = self.crop_padding(self.project_tetromino())
            return "\n".join("".join(row) for row in np.where(proj == 0, ".", proj.astype(str)))
        if self.render_mode == "rgb_array":
            # reference envs/tetris.py:364-373: get_rgb(_get_obs()) — board | queue strip / holder, the RgbObservation image — upscaled
            self._flush()
            img = self._engine.observe_as(N.OBS_RGB)[0]
            k = self.render_scaling_factor
            return np.kron(img, np.ones((k, k, 1), np.uint8))
        return None

    def close(self):
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None
