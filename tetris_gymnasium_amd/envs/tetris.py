"""`Tetris`: the reference's Gymnasium environment surface (tetris_gymnasium/envs/tetris.py:36-708) on top of
the HIP engine (rule-set NUMPY, one env).  Same constructor kwargs, `reset`/`step` signatures, observation
dict, rewards, and the attributes the reference's wrappers and tests reach into (`board, x, y,
active_tetromino, queue, holder, randomizer, has_swapped, game_over, padding, width_padded, ...`).

The attributes are plain host values: `step()` pushes them to the device (so poking `env.x`, `env.board`
or `env.active_tetromino` between steps works like in the reference's tests), runs ONE engine transition,
and pulls the new state back.  This is the drop-in / debugging surface; throughput comes from
`TetrisVectorEnv` / `Engine`, where state never leaves the GPU.
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
from ..engine import Engine
from ..mappings.actions import ActionsMapping
from ..mappings.rewards import RewardsMapping


class _QueueView:
    """Read view of the device-side TetrominoQueue (reference components/tetromino_queue.py)."""

    def __init__(self, size):
        self.size = size
        self.queue = deque(maxlen=size)

    def get_queue(self):
        return list(self.queue)


class _HolderView:
    """Read view of the device-side TetrominoHolder (reference components/tetromino_holder.py)."""

    def __init__(self, size=1):
        self.size = size
        self.queue = deque(maxlen=size)

    def get_tetrominoes(self):
        return list(self.queue)


class _RandomizerView:
    """Read view of the device-side BagRandomizer (reference components/tetromino_randomizer.py:49-102)."""

    def __init__(self, size=7):
        self.size = size
        self.bag = np.arange(size, dtype=np.int8)
        self.index = 0
        self.pcg64_state = None  # tg_rng_rec as a NumPy record


@dataclass
class TetrisState:
    board: np.ndarray
    rec: np.ndarray
    rng: np.ndarray


class Tetris(EnvBase):
    metadata = {"render_modes": ["rgb_array", "ansi"], "render_fps": 1}

    BASE_PIXELS = [Pixel(0, [0, 0, 0]), Pixel(1, [128, 128, 128])]
    TETROMINOES = [Tetromino(i, list(COLORS[i]), make_tetromino(i, 0, 1).matrix) for i in range(7)]

    def __init__(self, render_mode=None, width=10, height=20, gravity=True, actions_mapping=ActionsMapping(),
                 rewards_mapping=RewardsMapping(), queue=None, holder=None, randomizer=None, base_pixels=None,
                 tetrominoes=None, render_upscale: int = 10, queue_size: int = 4, device: int = 0):
        if any(a is not None for a in (queue, holder, randomizer, base_pixels, tetrominoes)):
            raise NotImplementedError("custom queue/holder/randomizer/base_pixels/tetrominoes are not supported by the "
                                      "HIP engine (they are also broken in the reference: envs/tetris.py:114-144)")
        if actions_mapping != ActionsMapping() or rewards_mapping != RewardsMapping():
            raise NotImplementedError("the HIP engine implements the stock ActionsMapping / RewardsMapping")
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
        self._engine = Engine(1, self.width, self.height, ruleset=N.RULESET_NUMPY, gravity=gravity,
                              queue_size=queue_size, obs_mode=N.OBS_DICT, device=device)
        self._rng_rec = None
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
    def _push(self):
        e = self._engine
        _, rec, rng = e.get_state()
        idx, rot, val = identify(self.active_tetromino)
        rec["piece"], rec["rot"], rec["val"] = idx, rot, val
        rec["x"], rec["y"] = self.x, self.y
        rec["flags"] = (N.F_GAME_OVER if self.game_over else 0) | (N.F_HAS_SWAPPED if self.has_swapped else 0)
        held = self.holder.get_tetrominoes()
        if held:
            hi_, hr, hv = identify(held[0])
            rec["holder_piece"], rec["holder_rot"], rec["holder_val"] = hi_, hr, hv
        else:
            rec["holder_piece"] = N.NONE
        q = list(self.queue.queue)
        rec["queue"][0, : len(q)] = q
        rec["bag"][0] = self.randomizer.bag
        rec["bag_index"] = self.randomizer.index
        p = self.padding
        board = np.ascontiguousarray(self.board[: self.height, p: p + self.width], np.uint8)[None]
        e.set_state(board=board, rec=rec, rng=self.randomizer.pcg64_state)  # (tg_set_state re-packs TG_BUF_OBS itself)

    def _pull(self):
        board, rec, rng = self._engine.get_state()
        r = rec[0]
        self.board = self.create_board()
        p = self.padding
        self.board[: self.height, p: p + self.width] = board[0]
        self.active_tetromino = make_tetromino(int(r["piece"]), int(r["rot"]), int(r["val"]))
        self.x, self.y = int(r["x"]), int(r["y"])
        self.game_over = bool(r["flags"] & N.F_GAME_OVER)
        self.has_swapped = bool(r["flags"] & N.F_HAS_SWAPPED)
        self.queue.queue = deque((np.int8(v) for v in r["queue"][: self.queue.size]), maxlen=self.queue.size)
        self.holder.queue = deque(maxlen=1)
        if r["holder_piece"] != N.NONE:
            self.holder.queue.append(make_tetromino(int(r["holder_piece"]), int(r["holder_rot"]), int(r["holder_val"])))
        self.randomizer.bag = r["bag"].astype(np.int8)
        self.randomizer.index = int(r["bag_index"])
        self.randomizer.pcg64_state = rng.copy()

    def _obs(self) -> "dict[str, Any]":
        # TG_BUF_OBS is what the step kernel wrote (after a transition) or what tg_reset / tg_set_state re-packed
        o = self._engine.observation()
        return {k: v[0].copy() for k, v in o.items()}

    # ---- gymnasium API --------------------------------------------------------------------------
    def step(self, action):
        """reference envs/tetris.py:203-272"""
        assert self.action_space.contains(action), f"{action!r} ({type(action)}) invalid"
        self._push()
        rew, term, lines = self._engine.step(np.array([int(action)], np.int32))
        self._pull()
        reward = int(rew[0]) if float(rew[0]).is_integer() else float(rew[0])
        return self._obs(), reward, bool(term[0]), False, {"lines_cleared": int(lines[0])}

    def reset(self, *, seed: "int | None" = None, options: "dict[str, Any] | None" = None):
        """reference envs/tetris.py:274-307 (seed=None or 0 keeps the running generator)"""
        try:
            super().reset(seed=seed, options=options)
        except TypeError:
            pass
        self._engine.reset(None if not seed else np.array([int(seed)], np.uint64))
        self._pull()
        return self._obs(), {"lines_cleared": 0}

    # ---- helpers the reference's wrappers / tests call ------------------------------------------
    def create_board(self) -> np.ndarray:
        b = np.zeros((self.height, self.width), np.uint8)
        return np.pad(b, ((0, self.padding), (self.padding, self.padding)), constant_values=1)

    def get_tetromino_slices(self, tetromino, x, y):
        h, w = tetromino.matrix.shape
        return (slice(y, y + h), slice(x, x + w))

    def collision(self, tetromino, x, y) -> bool:
        sub = self.board[self.get_tetromino_slices(tetromino, x, y)]
        return bool(np.any(sub[tetromino.matrix > 0] > 0))

    def rotate(self, tetromino, clockwise=True):
        return Tetromino(tetromino.id, tetromino.color_rgb, np.rot90(tetromino.matrix, k=(1 if clockwise else -1)))

    def crop_padding(self, matrix):
        return matrix[0: -self.padding, self.padding: -self.padding]

    def project_tetromino(self, tetromino=None, x=None, y=None):
        tetromino = self.active_tetromino if tetromino is None else tetromino
        x = self.x if x is None else x
        y = self.y if y is None else y
        proj = self.board.copy()
        if self.collision(tetromino, x, y):
            return proj
        proj[self.get_tetromino_slices(tetromino, x, y)] += tetromino.matrix
        return proj

    def score(self, rows_cleared) -> int:
        return (rows_cleared ** 2) * self.width

    def get_state(self) -> TetrisState:
        """Clone the full state incl. the PCG64 generator (reference envs/tetris.py:694-708)."""
        self._push()
        b, r, g = self._engine.get_state()
        return TetrisState(b.copy(), r.copy(), g.copy())

    def set_state(self, state: TetrisState) -> None:
        self._engine.set_state(board=state.board, rec=state.rec, rng=state.rng)
        self._pull()

    def render(self):
        if self.render_mode == "ansi":
            proj = self.crop_padding(self.project_tetromino())
            return "\n".join("".join(row) for row in np.where(proj == 0, ".", proj.astype(str)))
        if self.render_mode == "rgb_array":
            colors = np.array([p.color_rgb for p in self.pixels], np.uint8)
            img = colors[self.crop_padding(self.project_tetromino())]
            k = self.render_scaling_factor
            return np.kron(img, np.ones((k, k, 1), np.uint8))
        return None

    def close(self):
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None
