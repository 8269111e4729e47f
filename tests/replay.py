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


# ---- hand-made ("poked") states: oracle/refgen/make_golden.py gen_poked ----
SNAP_KEYS = ("queue", "bag", "bag_index", "holder_piece", "has_swapped", "game_over")


def poked_upload(fx, board, rec, skip=None, reps=8, n=None):
    """Write the fixture's hand-made states (the first `n`, each replicated `reps` times: one copy per first action) into a backend's
    (board, rec) arrays as get_state() returned them behind reset(seeds): everything the reference's tests assign directly — playfield,
    piece, rotation, position, holder, queue, bag, bag index, has_swapped, game_over.  `skip` [n_states] bool: states left as reset made them."""
    n, Q = int(fx["n_states"]) if n is None else int(n), int(fx["queue_size"])
    keep = (np.ones(n, bool) if skip is None else ~np.asarray(skip, bool)[:n])
    rep = lambda a: np.repeat(np.asarray(a)[:n], reps, axis=0)  # noqa: E731
    m = rep(keep)
    board[m] = rep(crop(fx["st_board"]))[m]
    rec["piece"][m] = rep(fx["st_piece"])[m]
    rec["rot"][m] = rep(fx["st_rot"])[m]
    rec["x"][m] = rep(fx["st_x"])[m]
    rec["y"][m] = rep(fx["st_y"])[m]
    rec["val"][m] = rep(fx["st_piece"])[m] + 2
    rec["flags"][m] = (rep(fx["st_game_over"]) | (rep(fx["st_has_swapped"]) << 1))[m]
    hp = rep(fx["st_holder_piece"])
    rec["holder_piece"][m] = hp[m]
    rec["holder_rot"][m] = np.where(hp == 255, 0, rep(fx["st_holder_rot"]))[m]
    rec["holder_val"][m] = np.where(hp == 255, 0, hp + 2)[m]
    rec["bag_index"][m] = rep(fx["st_bag_index"])[m]
    rec["bag"][m] = rep(fx["st_bag"])[m]
    rec["queue"][m, :Q] = rep(fx["st_queue"])[m]
    return m


def _cmp_snapshot(fx, prefix, board, rec, sel, msg):
    """(board, rec) of the envs `sel` against the reference's snapshot arrays `<prefix>_*` of the fixture."""
    Q = int(fx["queue_size"])
    np.testing.assert_array_equal(board[sel], crop(fx[prefix + "_board"])[sel], err_msg=f"board {msg}")
    np.testing.assert_array_equal(rec["x"][sel], fx[prefix + "_x"][sel], err_msg=f"x {msg}")
    np.testing.assert_array_equal(rec["y"][sel], fx[prefix + "_y"][sel], err_msg=f"y {msg}")
    np.testing.assert_array_equal(rec["piece"][sel], fx[prefix + "_piece"][sel], err_msg=f"piece {msg}")
    np.testing.assert_array_equal(rec["queue"][sel][:, :Q], fx[prefix + "_queue"][sel], err_msg=f"queue {msg}")
    np.testing.assert_array_equal(rec["bag"][sel], fx[prefix + "_bag"][sel], err_msg=f"bag {msg}")
    np.testing.assert_array_equal(rec["bag_index"][sel], fx[prefix + "_bag_index"][sel], err_msg=f"bag_index {msg}")
    np.testing.assert_array_equal(rec["holder_piece"][sel], fx[prefix + "_holder_piece"][sel], err_msg=f"holder {msg}")
    np.testing.assert_array_equal(((rec["flags"] & 2) >> 1)[sel], fx[prefix + "_has_swapped"][sel], err_msg=f"has_swapped {msg}")
    np.testing.assert_array_equal((rec["flags"] & 1)[sel], fx[prefix + "_game_over"][sel], err_msg=f"game_over {msg}")
    for i in np.flatnonzero(sel):
        np.testing.assert_array_equal(matrix16(rec["piece"][i], rec["rot"][i], rec["val"][i]), fx[prefix + "_matrix"][i],
                                      err_msg=f"matrix env={i} {msg}")
        if rec["holder_piece"][i] != 255:
            np.testing.assert_array_equal(matrix16(rec["holder_piece"][i], rec["holder_rot"][i], rec["holder_val"][i]),
                                          fx[prefix + "_holder_matrix"][i], err_msg=f"holder matrix env={i} {msg}")


def replay_poked(be, fx, skip=None, upload=None, step=None, check_obs=True, rollout=None):
    """Drive a backend through a poked_* fixture: upload the hand-made states, take every state's 8 first actions and the 15
    random transitions behind them, compare every output with the reference's as long as its recording of that copy lasts
    (`valid`).  be: the backend interface of replay_trace, constructed for batch 8 * n_states and reset with poked_seeds(fx);
    upload(board, rec): how the state gets in (default be.set_state); step(t, actions) -> (reward, terminated, lines);
    rollout(actions[T, B]) -> (reward, terminated, lines)[T, B]: all T transitions in fused launches instead — the outputs of every
    transition are compared, the state where the reference's recording runs to the end."""
    T = int(fx["T"])
    board, rec, _ = be.get_state()
    m = poked_upload(fx, board, rec, skip)
    (upload or (lambda b, r: be.set_state(board=b, rec=r)))(board, rec)
    if check_obs:
        obs = be.observe_dict()
        for i in np.flatnonzero(m):
            assert obs_crc(obs, i) == int(fx["st_crc"][i // 8]), f"observation of the uploaded state, env {i} (state {i // 8})"
    valid = fx["valid"].astype(bool) & m[None, :]
    last_t = valid.sum(axis=0) - 1  # the transition behind which the reference's `last_*` snapshot was taken
    if rollout is not None:
        rew, term, lines = rollout(fx["actions"].astype(np.int8))
        for t in range(T):
            v = valid[t]
            np.testing.assert_array_equal(np.asarray(rew[t])[v].astype(np.int64), fx["reward"][t][v].astype(np.int64), err_msg=f"reward t={t}")
            np.testing.assert_array_equal(np.asarray(term[t])[v], fx["done"][t][v], err_msg=f"terminated t={t}")
            np.testing.assert_array_equal(np.asarray(lines[t])[v], fx["lines"][t][v], err_msg=f"lines t={t}")
        board, rec, _ = be.get_state()
        sel = valid[T - 1] & (last_t == T - 1)
        assert sel.sum() > sel.size // 8
        _cmp_snapshot(fx, "last", board, rec, sel, "behind the fused launch")
        return m
    for t in range(T):
        a = fx["actions"][t].astype(np.int32)
        rew, term, lines = (step(t, a) if step else be.step(a))
        v = valid[t]
        np.testing.assert_array_equal(np.asarray(rew)[v].astype(np.int64), fx["reward"][t][v].astype(np.int64), err_msg=f"reward t={t}")
        np.testing.assert_array_equal(np.asarray(term)[v], fx["done"][t][v], err_msg=f"terminated t={t}")
        np.testing.assert_array_equal(np.asarray(lines)[v], fx["lines"][t][v], err_msg=f"lines t={t}")
        if check_obs:
            obs = be.observe_dict()
            bad = [i for i in np.flatnonzero(v) if obs_crc(obs, i) != int(fx["crc"][t][i])]
            assert not bad, f"observation crc mismatch t={t} envs={bad[:8]} (states {[i // 8 for i in bad[:8]]}, actions {[i % 8 for i in bad[:8]]})"
        board, rec, _ = be.get_state()
        np.testing.assert_array_equal(rec["x"][v], fx["x"][t][v], err_msg=f"x t={t}")
        np.testing.assert_array_equal(rec["y"][v], fx["y"][t][v], err_msg=f"y t={t}")
        np.testing.assert_array_equal(rec["piece"][v], fx["piece"][t][v], err_msg=f"piece t={t}")
        if t == 0:
            _cmp_snapshot(fx, "first", board, rec, v, "behind the first transition")
        sel = v & (last_t == t)
        if sel.any():
            _cmp_snapshot(fx, "last", board, rec, sel, f"at the end of the recording (t={t})")
    return m


def poked_seeds(fx, reps=8, n=None):
    return np.repeat(fx["seeds"].astype(np.uint64)[:n], reps)


def replay_poked_grouped(be_grouped_step, get_state, set_state, fx, skip=None):
    """Every one of the grouped wrapper's 4W actions from every hand-made state (wrappers/grouped.py:209-269, terminate_on_illegal_action
    = True): the backend holds n_grouped x 4W envs — state i replicated 4W times — and takes action a in copy a.  be_grouped_step(actions)
    -> (reward, terminated, lines)."""
    ng, NA = fx["gs_reward"].shape
    board, rec, _ = get_state()
    m = poked_upload(fx, board, rec, skip, reps=NA, n=ng)
    set_state(board, rec)
    a = np.tile(np.arange(NA, dtype=np.int32), ng)
    rew, term, lines = be_grouped_step(a)
    flat = lambda k: fx[k].reshape(ng * NA, *fx[k].shape[2:])  # noqa: E731
    np.testing.assert_array_equal(np.asarray(rew)[m], flat("gs_reward").astype(np.float32)[m], err_msg="grouped reward")
    np.testing.assert_array_equal(np.asarray(term)[m], flat("gs_done")[m], err_msg="grouped terminated")
    np.testing.assert_array_equal(np.asarray(lines)[m], flat("gs_lines")[m], err_msg="grouped lines")
    board, rec, _ = get_state()
    legal = m & (np.repeat(fx["g_mask"], 1, axis=0).reshape(-1) == 1)  # (an illegal action returns before touching the env)
    np.testing.assert_array_equal(board[legal], crop(flat("gs_board"))[legal], err_msg="board behind a grouped action")
    np.testing.assert_array_equal(rec["x"][legal], flat("gs_x")[legal]); np.testing.assert_array_equal(rec["y"][legal], flat("gs_y")[legal])
    np.testing.assert_array_equal(rec["piece"][legal], flat("gs_piece")[legal])
    assert legal.sum() > ng and (~legal & m).sum() > 0 and int(flat("gs_lines")[m].sum()) > 0
    return m
