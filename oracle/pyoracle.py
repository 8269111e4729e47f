"""ctypes front-end of oracle/libtg_oracle.so — TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg.  The
product package (tetris_gymnasium_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TG_ORACLE_LIB: another build of the same source (the AddressSanitizer / UBSan build of tests/test_oracle_sanitize.py)
_LIB_PATH = os.environ.get("TG_ORACLE_LIB") or os.path.join(_HERE, "libtg_oracle.so")

RULESET_NUMPY, RULESET_FUNCTIONAL = 0, 1
OBS_NONE, OBS_BOARD, OBS_DICT, OBS_FN, OBS_RGB = 0, 1, 2, 3, 4
AUTORESET_NONE, AUTORESET_SAME_STEP, AUTORESET_NEXT_STEP = 0, 1, 2
NONE = 0xFF
F_GAME_OVER, F_HAS_SWAPPED, F_PENDING_RESET = 1, 2, 4

ENV_REC = np.dtype(
    [
        ("piece", "u1"), ("rot", "u1"), ("x", "u1"), ("y", "u1"), ("val", "u1"), ("flags", "u1"),
        ("holder_piece", "u1"), ("holder_rot", "u1"), ("holder_val", "u1"), ("bag_index", "u1"),
        ("bag", "u1", (7,)), ("queue", "u1", (7,)), ("score", "<f4"), ("reserved", "<u4"),
    ]
)
RNG_REC = np.dtype(
    [
        ("state_lo", "<u8"), ("state_hi", "<u8"), ("inc_lo", "<u8"), ("inc_hi", "<u8"),
        ("uinteger", "<u4"), ("has_uint32", "<u4"), ("key", "<u4", (2,)),
    ]
)
assert ENV_REC.itemsize == 32 and RNG_REC.itemsize == 48


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc) if the shared object is missing or stale."""
    src = os.path.join(_HERE, "tg_oracle.c")
    hdrs = [os.path.join(_HERE, "tg_oracle.h"), os.path.join(_HERE, "..", "include", "tetris_gym_hip.h")]
    if os.environ.get("TG_ORACLE_LIB"):
        return _LIB_PATH
    stale = force or not os.path.exists(_LIB_PATH)
    if not stale:
        t = os.path.getmtime(_LIB_PATH)
        stale = any(os.path.exists(p) and os.path.getmtime(p) > t for p in [src] + hdrs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtg_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
        L.tgo_create.restype = vp
        L.tgo_create.argtypes = [i32] * 8
        L.tgo_destroy.argtypes = [vp]
        L.tgo_set_mappings.argtypes = [vp, vp, vp]
        L.tgo_last_rewards.argtypes = [vp, vp]
        L.tgo_reset.argtypes = [vp, vp, vp]
        L.tgo_step.argtypes = [vp, vp, vp, vp, vp, i32, i32]
        L.tgo_grouped_step.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32]
        L.tgo_grouped_enumerate.argtypes = [vp, vp, vp, i32]
        L.tgo_observe.argtypes = [vp, i32, vp, i32]
        L.tgo_grouped_features.argtypes = [vp, vp, i32]
        L.tgo_features.argtypes = [vp, vp, i32]
        L.tgo_get_state.argtypes = [vp, vp, vp, vp]
        L.tgo_set_state.argtypes = [vp, vp, vp, vp]
        L.tgo_get_padded_boards.argtypes = [vp, vp]
        L.tgo_set_padded_boards.argtypes = [vp, vp]
        L.tgo_seed_pcg64.argtypes = [u64, vp]
        L.tgo_pcg64_next32.restype = C.c_uint32
        L.tgo_pcg64_next32.argtypes = [vp]
        L.tgo_pcg64_shuffle7.argtypes = [vp, vp]
        L.tgo_threefry2x32.argtypes = [vp, vp, vp]
        L.tgo_jax_split.argtypes = [vp, vp, vp]
        L.tgo_jax_permutation7.argtypes = [vp, vp]
        L.tgo_jax_permutation.argtypes = [vp, i32, vp]
        L.tgo_set_threefry_partitionable.argtypes = [i32]
        L.tgo_fn_score.restype = i32
        L.tgo_fn_score.argtypes = [i32]
        L.tgo_fn_clear_rows.restype = i32
        L.tgo_fn_clear_rows.argtypes = [vp, i32, i32]
        L.tgo_np_score.restype = i32
        L.tgo_np_score.argtypes = [i32, i32]
        L.tgo_np_matrix.restype = i32
        L.tgo_np_matrix.argtypes = [i32, i32, vp]
        L.tgo_fn_matrix.argtypes = [i32, i32, vp]
        L.tgo_feature_vector.argtypes = [vp, vp, i32, i32, i32, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleBatch:
    """A batch of CPU oracle envs with the same call surface as the engine's `Engine` class."""

    def __init__(self, batch, width=10, height=20, ruleset=RULESET_NUMPY, gravity=True, queue_size=None,
                 padding=4, env_offset=0, nthreads=1, action_ids=None, rewards=None):
        """action_ids / rewards: the field values of an ActionsMapping (8) / RewardsMapping (4) in declaration order
        (mappings/actions.py:12-19, mappings/rewards.py:12-15); None = the stock dataclasses."""
        self.L = lib()
        if queue_size is None:
            queue_size = 4 if ruleset == RULESET_NUMPY else 7
        self.B, self.W, self.H, self.P, self.Q = batch, width, height, padding, queue_size
        self.Wp, self.Hp = width + 2 * padding, height + padding
        self.ruleset, self.nthreads = ruleset, nthreads
        self.h = self.L.tgo_create(ruleset, width, height, padding, queue_size, int(bool(gravity)), batch,
                                   env_offset)
        if not self.h:
            raise ValueError("tgo_create rejected the configuration")
        if action_ids is not None or rewards is not None:
            a = None if action_ids is None else np.ascontiguousarray(action_ids, dtype=np.int32)
            r = None if rewards is None else np.ascontiguousarray(rewards, dtype=np.float64)
            assert (a is None or a.shape == (8,)) and (r is None or r.shape == (4,))
            self.L.tgo_set_mappings(self.h, _p(a), _p(r))

    def last_rewards(self):
        """The rewards of the last step as Python computes them (int score + float alife in double precision)."""
        out = np.empty(self.B, np.float64)
        self.L.tgo_last_rewards(self.h, _p(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.L.tgo_destroy(self.h)
            self.h = None

    def reset(self, seeds=None, mask=None):
        s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.tgo_reset(self.h, _p(s), _p(m))

    def step(self, actions, autoreset=AUTORESET_NONE):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        rew = np.empty(self.B, np.float32)
        term = np.empty(self.B, np.uint8)
        lines = np.empty(self.B, np.uint8)
        self.L.tgo_step(self.h, _p(a), _p(rew), _p(term), _p(lines), autoreset, self.nthreads)
        return rew, term, lines

    def grouped_step(self, actions, terminate_on_illegal=True, autoreset=AUTORESET_NONE):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        rew = np.empty(self.B, np.float32)
        term = np.empty(self.B, np.uint8)
        lines = np.empty(self.B, np.uint8)
        ill = np.empty(self.B, np.uint8)
        self.L.tgo_grouped_step(self.h, _p(a), _p(rew), _p(term), _p(lines), _p(ill),
                                int(terminate_on_illegal), autoreset, self.nthreads)
        return rew, term, lines, ill

    def grouped_enumerate(self, boards=False):
        mask = np.empty((self.B, 4 * self.W), np.uint8)
        bo = np.empty((self.B, 4 * self.W, self.Hp, self.Wp), np.uint8) if boards else None
        self.L.tgo_grouped_enumerate(self.h, _p(mask), _p(bo), self.nthreads)
        return (mask, bo) if boards else mask

    def grouped_features(self):
        out = np.empty((self.B, 4 * self.W, self.W + 3), np.uint8)
        self.L.tgo_grouped_features(self.h, _p(out), self.nthreads)
        return out

    def features(self):
        out = np.empty((self.B, self.W + 3), np.uint8)
        self.L.tgo_features(self.h, _p(out), self.nthreads)
        return out

    def observe(self, mode):
        if mode == OBS_BOARD:
            out = np.empty((self.B, self.H, self.W), np.uint8)
            self.L.tgo_observe(self.h, mode, _p(out), self.nthreads)
            return out
        if mode == OBS_FN:
            out = np.empty((self.B, self.H, self.W), np.int8)
            self.L.tgo_observe(self.h, mode, _p(out), self.nthreads)
            return out
        if mode == OBS_DICT:
            cells, pp = self.Hp * self.Wp, self.P * self.P
            flat = np.empty(self.B * (2 * cells + pp + pp * self.Q), np.uint8)
            self.L.tgo_observe(self.h, mode, _p(flat), self.nthreads)
            return split_dict_obs(flat, self.B, self.Hp, self.Wp, self.P, self.Q)
        if mode == OBS_RGB:
            out = np.empty((self.B, self.Hp, self.Wp + self.P * max(1, self.Q), 3), np.uint8)
            self.L.tgo_observe(self.h, mode, _p(out), self.nthreads)
            return out
        raise ValueError(mode)

    def get_state(self):
        board = np.empty((self.B, self.H, self.W), np.uint8)
        rec = np.empty(self.B, ENV_REC)
        rng = np.empty(self.B, RNG_REC)
        self.L.tgo_get_state(self.h, _p(board), _p(rec), _p(rng))
        return board, rec, rng

    def set_state(self, board=None, rec=None, rng=None):
        b = None if board is None else np.ascontiguousarray(board, dtype=np.uint8)
        r = None if rec is None else np.ascontiguousarray(rec, dtype=ENV_REC)
        g = None if rng is None else np.ascontiguousarray(rng, dtype=RNG_REC)
        self.L.tgo_set_state(self.h, _p(b), _p(r), _p(g))

    def padded_boards(self):
        out = np.empty((self.B, self.Hp, self.Wp), np.uint8)
        self.L.tgo_get_padded_boards(self.h, _p(out))
        return out

    def set_padded_boards(self, boards):
        b = np.ascontiguousarray(boards, dtype=np.uint8)
        assert b.shape == (self.B, self.Hp, self.Wp)
        self.L.tgo_set_padded_boards(self.h, _p(b))


def split_dict_obs(flat, B, Hp, Wp, P, Q):
    """Split the TG_OBS_DICT buffer layout into the reference's four observation arrays."""
    cells, pp = Hp * Wp, P * P
    o = 0
    board = flat[o:o + B * cells].reshape(B, Hp, Wp); o += B * cells
    mask = flat[o:o + B * cells].reshape(B, Hp, Wp); o += B * cells
    holder = flat[o:o + B * pp].reshape(B, P, P); o += B * pp
    queue = flat[o:o + B * pp * Q].reshape(B, P, P * Q)
    return {"board": board, "active_tetromino_mask": mask, "holder": holder, "queue": queue}


def seed_pcg64(seed: int) -> np.ndarray:
    r = np.zeros(1, RNG_REC)
    lib().tgo_seed_pcg64(int(seed), _p(r))
    return r


def feature_vector(board, mask, width, height, padding=4):
    out = np.zeros(width + 3, np.uint8)
    b = np.ascontiguousarray(board, np.uint8)
    m = np.ascontiguousarray(mask, np.uint8)
    lib().tgo_feature_vector(_p(b), _p(m), width, height, padding, _p(out))
    return out
