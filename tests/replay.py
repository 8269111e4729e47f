"""Shared trace-replay checker: drives a backend (CPU oracle or HIP engine) through a golden trace
generated from the reference NumPy env (oracle/refgen/make_golden.py) and compares every output.

A backend exposes: reset(seeds, mask), step(actions) -> (reward, terminated, lines), observe_dict() ->
dict of 4 arrays, get_state() -> (board[B,H,W], rec, rng), set_state(board=..., rec=..., rng=...).
"""
from __future__ import annotations

import os
import zlib

import numpy as np

NATIVE = {
    0: np.array([[0, 0, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]], np.uint8),
    1: np.array([[1, 1], [1, 1]], np.uint8),
    2: np.array([[0, 1, 0], [1, 1, 1], [0, 0, 0]], np.uint8),
    3: np.array([[0, 1, 1], [1, 1, 0], [0, 0, 0]], np.uint8),
    4: np.array([[1, 1, 0], [0, 1, 1], [0, 0, 0]], np.uint8),
    5: np.array([[1, 0, 0], [1, 1, 1], [0, 0, 0]], np.uint8),
    6: np.array([[0, 0, 1], [1, 1, 1], [0, 0, 0]], np.uint8),
}


def matrix16(piece: int, rot: int, val: int) -> np.ndarray:
    m = np.rot90(NATIVE[int(piece)] * np.uint8(val), k=int(rot))
    out = np.zeros((4, 4), np.uint8)
    out[: m.shape[0], : m.shape[1]] = m
    return out.reshape(16)


def obs_crc(obs, i) -> int:
    c = 0
    for k in ("board", "active_tetromino_mask", "holder", "queue"):
        c = zlib.crc32(np.ascontiguousarray(obs[k][i]).tobytes(), c)
    return c


def crop(boards, P=4):
    return np.ascontiguousarray(boards[:, :-P, P:-P])


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def replay_trace(make_backend, fx, max_steps=None, check_state_every=1):
    W, H, B, T = int(fx["width"]), int(fx["height"]), int(fx["B"]), int(fx["T"])
    if max_steps:
        T = min(T, max_steps)
    mapped = "action_ids" in fx.files  # fixtures made with a non-stock ActionsMapping / RewardsMapping (envs/tetris.py:83-84)
    kw = dict(action_ids=fx["action_ids"], rewards=fx["rewards_mapping"]) if mapped else {}
    be = make_backend(batch=B, width=W, height=H, gravity=bool(fx["gravity"]), queue_size=int(fx["queue_size"]), **kw)
    be.reset(seeds=1 + np.arange(B, dtype=np.uint64))
    init = fx["init_boards"] if "init_boards" in fx.files else None
    if init is not None:
        be.set_state(board=crop(init))
    obs = be.observe_dict()
    for i in range(B):
        assert obs_crc(obs, i) == int(fx["first_obs_crc"][i]), f"reset obs mismatch env {i}"
    snap_steps = list(fx["snap_steps"])
    for t in range(T):
        rew, term, lines = be.step(fx["actions"][t].astype(np.int32))
        if mapped:  # the Python value (int score + the mapping's floats, double precision) and its float32 rounding
            np.testing.assert_array_equal(be.rewards_f64(), fx["reward_f64"][t], err_msg=f"reward (f64) t={t}")
            np.testing.assert_array_equal(rew, fx["reward_f64"][t].astype(np.float32), err_msg=f"reward (f32) t={t}")
        else:
            np.testing.assert_array_equal(rew.astype(np.int64), fx["reward"][t].astype(np.int64), err_msg=f"reward t={t}")
        np.testing.assert_array_equal(term, fx["done"][t], err_msg=f"terminated t={t}")
        np.testing.assert_array_equal(lines, fx["lines"][t], err_msg=f"lines t={t}")
        obs = be.observe_dict()
        got = np.array([obs_crc(obs, i) for i in range(B)], np.uint32)
        bad = np.flatnonzero(got != fx["crc"][t])
        assert bad.size == 0, f"observation crc mismatch t={t} envs={bad[:8]}"
        if t % check_state_every == 0 or t in snap_steps:
            board, rec, _ = be.get_state()
            np.testing.assert_array_equal(rec["x"], fx["x"][t], err_msg=f"x t={t}")
            np.testing.assert_array_equal(rec["y"], fx["y"][t], err_msg=f"y t={t}")
            np.testing.assert_array_equal(rec["piece"], fx["piece"][t], err_msg=f"piece t={t}")
        if t in snap_steps:
            si = snap_steps.index(t)
            np.testing.assert_array_equal(board, crop(fx["snap_board"][si]), err_msg=f"board t={t}")
            Q = int(fx["queue_size"])
            np.testing.assert_array_equal(rec["queue"][:, :Q], fx["snap_queue"][si], err_msg=f"queue t={t}")
            np.testing.assert_array_equal(rec["bag"], fx["snap_bag"][si])
            np.testing.assert_array_equal(rec["bag_index"], fx["snap_bag_index"][si])
            np.testing.assert_array_equal(rec["holder_piece"], fx["snap_holder_piece"][si])
            np.testing.assert_array_equal((rec["flags"] & 2) >> 1, fx["snap_has_swapped"][si])
            np.testing.assert_array_equal(rec["flags"] & 1, fx["snap_game_over"][si])
            for i in range(B):
                np.testing.assert_array_equal(matrix16(rec["piece"][i], rec["rot"][i], rec["val"][i]),
                                              fx["snap_matrix"][si][i], err_msg=f"matrix t={t} env={i}")
                if rec["holder_piece"][i] != 255:
                    np.testing.assert_array_equal(
                        matrix16(rec["holder_piece"][i], rec["holder_rot"][i], rec["holder_val"][i]),
                        fx["snap_holder_matrix"][si][i])
        if term.any():
            be.reset(seeds=None, mask=term)
            if init is not None:
                board, _, _ = be.get_state()
                board[term.astype(bool)] = crop(init)[term.astype(bool)]
                be.set_state(board=board)
            obs = be.observe_dict()
            for i in np.flatnonzero(term):
                assert obs_crc(obs, i) == int(fx["reset_crc"][t][i]), f"post-reset obs mismatch t={t} env={i}"
    return be
