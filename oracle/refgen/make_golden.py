"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference NumPy env.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); the GPU box sees only
the committed .npz outputs.  Nothing of the reference's source is stored — inputs (seeds, actions,
initial boards) and the reference's outputs (rewards, dones, lines, boards, observation checksums).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refgen.make_golden [--only NAME]

Fixtures (all NumPy rule-set, i.e. envs/tetris.py — the graded oracle of BASELINE.json):
  pcg64_streams.npz      SeedSequence→PCG64 initial states + first 70 bag pieces per seed
  trace_<cfg>.npz        random-policy traces, harness reset() on done
  clears_<cfg>.npz       line-clear-dense traces from pre-filled boards
  grouped_<cfg>.npz      grouped wrapper: legal masks, placement boards, traces (grouped_clears_*: on dense stacks)
  features.npz           FeatureVectorObservation outputs on trace states
  grouped_features_*.npz per-placement feature tensors [4W][W+3] (grouped wrapper + FeatureVectorObservation)
  ref_tests_grouped.npz  the reference tests' own mock board / CSV board / 40-entry mask (data only)
  rgb_<cfg>.npz          RgbObservation frames (wrappers/observation.py:38-74) along random traces incl. hold swaps
  grouped_rgb_10x20.npz  the grouped wrapper with observation_wrappers=[RgbObservation]: one RGB image per placement
  poked_<cfg>.npz        HAND-MADE states in the style of the reference's own tests (tests/test_base_env/**: board / x / y /
                         active_tetromino / holder / queue / has_swapped / game_over assigned directly) x each of the 8 actions, then 15
                         random transitions; the grouped wrapper's mask / placement boards / features and the RGB frame of the same states
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import zlib

import numpy as np

from oracle.refgen.standins import import_reference

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def obs_crc(obs) -> int:
    c = 0
    for k in ("board", "active_tetromino_mask", "holder", "queue"):
        c = zlib.crc32(np.ascontiguousarray(obs[k], dtype=np.uint8).tobytes(), c)
    return c


def matrix16(t) -> np.ndarray:
    m = np.zeros((4, 4), np.uint8)
    n = t.matrix.shape[0]
    m[:n, :n] = t.matrix
    return m.reshape(16)


def mapping_kwargs(action_ids=None, rewards=None):
    """Tetris(actions_mapping=..., rewards_mapping=...) from field values in declaration order (mappings/actions.py:12-19,
    mappings/rewards.py:12-15)"""
    from tetris_gymnasium.mappings.actions import ActionsMapping
    from tetris_gymnasium.mappings.rewards import RewardsMapping

    kw = {}
    if action_ids is not None:
        kw["actions_mapping"] = ActionsMapping(*[int(v) for v in action_ids])
    if rewards is not None:
        kw["rewards_mapping"] = RewardsMapping(*rewards)
    return kw


def make_env(ref, width, height, gravity, queue_size, action_ids=None, rewards=None):
    env = ref.Tetris(width=width, height=height, gravity=gravity, **mapping_kwargs(action_ids, rewards))
    if queue_size != 4:
        # the reference's queue=/holder= kwargs are broken (envs/tetris.py:139-144 leave the attribute
        # unset); config 5 patches the queue after construction
        from tetris_gymnasium.components.tetromino_queue import TetrominoQueue

        env.queue = TetrominoQueue(env.randomizer, queue_size)
    return env


def snapshot(env):
    hold = env.holder.get_tetrominoes()
    return dict(
        board=env.board.copy(),
        x=env.x, y=env.y,
        piece=env.active_tetromino.id - 2,
        matrix=matrix16(env.active_tetromino),
        queue=np.array(list(env.queue.queue), np.uint8),
        holder_piece=(hold[0].id - 2) if hold else 255,
        holder_matrix=matrix16(hold[0]) if hold else np.zeros(16, np.uint8),
        bag=env.randomizer.bag.astype(np.uint8).copy(),
        bag_index=env.randomizer.index,
        has_swapped=int(env.has_swapped),
        game_over=int(env.game_over),
    )


def gen_pcg64_streams():
    """numpy SeedSequence → PCG64 state, and the BagRandomizer stream (reference component)."""
    ref = import_reference()
    from tetris_gymnasium.components.tetromino_randomizer import BagRandomizer

    seeds = np.array([1, 2, 3, 42, 123, 2**31 - 1, 2**32 + 5, 2**63 + 11, 65536, 524288], dtype=np.uint64)
    states = np.zeros((len(seeds), 4), np.uint64)  # state_lo, state_hi, inc_lo, inc_hi
    streams = np.zeros((len(seeds), 70), np.uint8)
    next32 = np.zeros((len(seeds), 9), np.uint32)
    for k, s in enumerate(seeds):
        bg = np.random.PCG64(np.random.SeedSequence(int(s)))
        st = bg.state["state"]
        states[k] = [st["state"] & (2**64 - 1), st["state"] >> 64, st["inc"] & (2**64 - 1), st["inc"] >> 64]
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(s))))
        # 9 consecutive 32-bit draws (odd count exercises the buffered high half)
        next32[k] = g.integers(0, 2**32, size=9, dtype=np.uint32)
        r = BagRandomizer(7)
        r.reset(seed=int(s))
        streams[k] = [r.get_next_tetromino() for _ in range(70)]
    np.savez_compressed(os.path.join(OUT, "pcg64_streams.npz"), seeds=seeds, states=states, streams=streams,
                        next32=next32, numpy_version=np.__version__)
    print("pcg64_streams: seed 42 stream", streams[3][:14])


class SeekPolicy:
    """Scripted controller used only to make line clears frequent: per piece pick a target rotation and a
    target column inside the stack's gap, rotate, walk there, hard-drop.  Its decisions are stored in the
    fixture as plain actions, so tests replay actions only."""

    def __init__(self, rng, gap_lo, gap_hi, padding=4, action_ids=None):
        self.rng, self.lo, self.hi, self.p = rng, gap_lo, gap_hi, padding
        self.plan = False
        self.ids = list(range(8)) if action_ids is None else [int(v) for v in action_ids]  # the env's ActionsMapping

    def act(self, env):
        if not self.plan:
            self.rots = int(self.rng.integers(0, 4))
            self.tx = int(self.rng.integers(self.lo + self.p - 1, self.hi + self.p + 1))
            self.moves = 0
            self.plan = True
        if self.rots > 0:
            self.rots -= 1
            return self.ids[3]  # rotate_clockwise
        if env.x != self.tx and self.moves < 14:
            self.moves += 1
            return self.ids[0] if env.x > self.tx else self.ids[1]  # move_left / move_right
        self.plan = False
        return self.ids[5] if self.rng.random() < 0.9 else int(self.rng.integers(0, 8))  # hard_drop


def run_trace(ref, name, width, height, gravity, queue_size, B, T, action_seed, n_actions=8,
              snap_every=64, init_boards=None, action_p=None, seek=None, action_ids=None, rewards=None):
    rng = np.random.default_rng(action_seed)
    if action_p is None:
        actions = rng.integers(0, n_actions, size=(T, B)).astype(np.int8)
    else:
        actions = rng.choice(n_actions, size=(T, B), p=action_p).astype(np.int8)
    reward = np.zeros((T, B), np.int16)
    reward_f64 = np.zeros((T, B), np.float64)  # the Python value (int score + RewardsMapping floats) for non-stock mappings
    done = np.zeros((T, B), np.uint8)
    lines = np.zeros((T, B), np.uint8)
    crc = np.zeros((T, B), np.uint32)
    reset_crc = np.zeros((T, B), np.uint32)
    xs = np.zeros((T, B), np.uint8)
    ys = np.zeros((T, B), np.uint8)
    pieces = np.zeros((T, B), np.uint8)
    snaps = {}
    snap_steps = [t for t in range(T) if (t % snap_every) == snap_every - 1]
    Hp, Wp = height + 4, width + 8
    snap_board = np.zeros((len(snap_steps), B, Hp, Wp), np.uint8)
    snap_rec = {k: [] for k in ("x", "y", "piece", "matrix", "queue", "holder_piece", "holder_matrix", "bag",
                                "bag_index", "has_swapped", "game_over")}
    for k in snap_rec:
        snap_rec[k] = [[None] * B for _ in snap_steps]
    first_obs_crc = np.zeros(B, np.uint32)
    t0 = time.time()
    for i in range(B):
        env = make_env(ref, width, height, gravity, queue_size, action_ids, rewards)
        obs, _ = env.reset(seed=1 + i)
        if init_boards is not None:
            env.board = init_boards[i].copy()
            obs = env._get_obs()
        first_obs_crc[i] = obs_crc(obs)
        pol = SeekPolicy(rng, *seek[i], action_ids=action_ids) if seek is not None else None
        for t in range(T):
            if pol is not None:
                actions[t, i] = pol.act(env)
            obs, r, d, _, info = env.step(int(actions[t, i]))
            assert rewards is not None or r == int(r)
            reward_f64[t, i] = float(r)
            reward[t, i], done[t, i], lines[t, i] = (int(r) if r == int(r) else 0), int(d), int(info["lines_cleared"])
            crc[t, i] = obs_crc(obs)
            xs[t, i], ys[t, i], pieces[t, i] = env.x, env.y, env.active_tetromino.id - 2
            if (t % snap_every) == snap_every - 1:
                si = snap_steps.index(t)
                s = snapshot(env)
                snap_board[si, i] = s["board"]
                for k in snap_rec:
                    snap_rec[k][si][i] = s[k]
            if d:
                obs, _ = env.reset()
                if init_boards is not None:  # re-inject the dense stack after every harness reset
                    env.board = init_boards[i].copy()
                    obs = env._get_obs()
                reset_crc[t, i] = obs_crc(obs)
                if pol is not None:
                    pol.plan = False
    out = dict(
        width=width, height=height, gravity=int(gravity), queue_size=queue_size, B=B, T=T,
        actions=actions, reward=reward, done=done, lines=lines, crc=crc, reset_crc=reset_crc,
        x=xs, y=ys, piece=pieces, first_obs_crc=first_obs_crc,
        snap_steps=np.array(snap_steps), snap_board=snap_board,
    )
    for k, v in snap_rec.items():
        out["snap_" + k] = np.array(v).astype(np.uint8)
    if init_boards is not None:
        out["init_boards"] = init_boards
    if action_ids is not None or rewards is not None:
        out["reward_f64"] = reward_f64
        out["action_ids"] = np.array(list(range(8)) if action_ids is None else action_ids, np.int32)
        out["rewards_mapping"] = np.array([1, 1, 0, -0.1] if rewards is None else rewards, np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {B*T} steps in {time.time()-t0:.1f}s  episodes={int(done.sum())}  "
          f"lock-steps={(reward>0).sum()}  lines total={int(lines.sum())} max={int(lines.max())}")


def dense_boards(width, height, B, seed):
    """Nearly-full stacks: rows [H-k, H) filled with cell values 2..8 except a gap of 1-4 columns (the
    same columns in every row, so dropped pieces complete several rows at once).  Returns boards, gaps."""
    rng = np.random.default_rng(seed)
    Hp, Wp, P = height + 4, width + 8, 4
    boards = np.ones((B, Hp, Wp), np.uint8)
    boards[:, :height, P:P + width] = 0
    gaps = []
    for i in range(B):
        k = int(rng.integers(3, max(4, height // 2 - 2)))
        gw = int(rng.integers(1, 5))
        g = int(rng.integers(0, width - gw + 1))
        gaps.append((g, g + gw - 1))
        for r in range(height - k, height):
            row = rng.integers(2, 9, size=width).astype(np.uint8)
            row[g:g + gw] = 0
            if rng.random() < 0.15:
                row[int(rng.integers(0, width))] = 0
            boards[i, r, P:P + width] = row
    return boards, gaps


def gen_clears(ref, name, width, height, queue_size, B, T, seed, board_seed, action_ids=None, rewards=None):
    boards, gaps = dense_boards(width, height, B, board_seed)
    run_trace(ref, name, width, height, True, queue_size, B, T, seed, snap_every=32, init_boards=boards,
              seek=gaps, action_ids=action_ids, rewards=rewards)


def gen_grouped(ref, name, width, height, queue_size, B, T, seed, n_enum, init_boards=None, action_ids=None, rewards=None,
                terminate_on_illegal=True, p_any=0.03):
    """Grouped wrapper traces (random LEGAL action, like examples/train_lin_grouped.py's epsilon branch)
    plus full enumerations (mask + 4W boards) on mid-game states."""
    rng = np.random.default_rng(seed)
    NA = 4 * width
    Hp, Wp = height + 4, width + 8
    actions = np.zeros((T, B), np.int16)
    reward = np.zeros((T, B), np.float32)
    reward_f64 = np.zeros((T, B), np.float64)
    done = np.zeros((T, B), np.uint8)
    lines = np.zeros((T, B), np.uint8)
    masks = np.zeros((T, B, NA), np.uint8)
    board_after = np.zeros((T, B, Hp, Wp), np.uint8)
    enum_state_board = np.zeros((n_enum, Hp, Wp), np.uint8)
    enum_piece = np.zeros(n_enum, np.uint8)
    enum_matrix = np.zeros((n_enum, 16), np.uint8)
    enum_mask = np.zeros((n_enum, NA), np.uint8)
    enum_boards = np.zeros((n_enum, NA, Hp, Wp), np.uint8)
    ne = 0
    reset_mask = np.zeros((B, NA), np.uint8)
    t0 = time.time()
    for i in range(B):
        base = make_env(ref, width, height, True, queue_size, action_ids, rewards)
        env = ref.GroupedActionsObservations(base, terminate_on_illegal_action=terminate_on_illegal)
        obs, info = env.reset(seed=1 + i)
        if init_boards is not None:
            base.board = init_boards[i].copy()
            env.observation(base._get_obs())  # refresh the cached legal mask
        reset_mask[i] = env.legal_actions_mask
        for t in range(T):
            mask = env.legal_actions_mask
            legal = np.flatnonzero(mask == 1)
            # mostly legal actions, sometimes any action (exercises the illegal → terminate branch)
            if len(legal) == 0 or rng.random() < p_any:
                a = int(rng.integers(0, NA))
            else:
                a = int(rng.choice(legal))
            actions[t, i] = a
            obs, r, d, _, info = env.step(a)
            reward[t, i], done[t, i], lines[t, i] = r, int(d), int(info["lines_cleared"])
            reward_f64[t, i] = float(r)
            masks[t, i] = info["action_mask"]
            board_after[t, i] = base.board
            if ne < n_enum and not d and rng.random() < 0.08:
                enum_state_board[ne] = base.board
                enum_piece[ne] = base.active_tetromino.id - 2
                enum_matrix[ne] = matrix16(base.active_tetromino)
                enum_mask[ne] = env.legal_actions_mask
                enum_boards[ne] = obs.astype(np.uint8)
                ne += 1
            if d:
                obs, info = env.reset()
                if init_boards is not None:
                    base.board = init_boards[i].copy()
                    env.observation(base._get_obs())
                masks[t, i] = env.legal_actions_mask  # mask the next action is drawn from
    extra = {}
    if action_ids is not None or rewards is not None:
        extra = dict(reward_f64=reward_f64, terminate_on_illegal=int(terminate_on_illegal),
                     action_ids=np.array(list(range(8)) if action_ids is None else action_ids, np.int32),
                     rewards_mapping=np.array([1, 1, 0, -0.1] if rewards is None else rewards, np.float64))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), width=width, height=height, queue_size=queue_size, B=B, T=T,
        actions=actions, reward=reward, done=done, lines=lines, masks=masks, board_after=board_after,
        reset_mask=reset_mask, **extra, init_boards=(init_boards if init_boards is not None else np.zeros(0, np.uint8)),
        enum_state_board=enum_state_board[:ne], enum_piece=enum_piece[:ne],
        enum_matrix=enum_matrix[:ne], enum_mask=enum_mask[:ne], enum_boards=enum_boards[:ne])
    print(f"{name}: {B*T} grouped steps in {time.time()-t0:.1f}s episodes={int(done.sum())} "
          f"lines={int(lines.sum())} enums={ne} illegal={(reward<0).sum()}")


def gen_grouped_features(ref, name, width, height, queue_size, n_states, seed, dense=False, prefull=False):
    """GroupedActionsObservations(env, observation_wrappers=[FeatureVectorObservation(env)]) — the observation
    examples/train_lin_grouped.py trains on: for each of the 4W placements the 13-entry feature vector of the
    resulting board (wrappers/grouped.py:124-207 + wrappers/observation.py:236-278), plus info["board"].
    prefull: every recorded state gets one to three rows that are ALREADY full (a board only a caller can provide — during play a
    full row never survives its lock): `clear_filled_rows` of every placement takes them too (wrappers/grouped.py:174,
    envs/tetris.py:493); the placement boards of the wrapper WITHOUT observation wrappers are recorded beside the features."""
    rng = np.random.default_rng(seed)
    NA, Hp, Wp = 4 * width, height + 4, width + 8
    boards = np.zeros((n_states, Hp, Wp), np.uint8)
    pieces = np.zeros(n_states, np.uint8)
    mats = np.zeros((n_states, 16), np.uint8)
    feats = np.zeros((n_states, NA, width + 3), np.uint8)
    cur_feats = np.zeros((n_states, width + 3), np.uint8)
    masks = np.zeros((n_states, NA), np.uint8)
    base = make_env(ref, width, height, True, queue_size)
    env = ref.GroupedActionsObservations(base, observation_wrappers=[ref.FeatureVectorObservation(base)])
    plain = ref.GroupedActionsObservations(base) if prefull else None
    pboards = np.zeros((n_states, NA, Hp, Wp), np.uint8) if prefull else None
    dense_boards_, _ = dense_boards(width, height, 64, seed + 1)
    obs, info = env.reset(seed=seed)
    k = 0
    while k < n_states:
        if dense and rng.random() < 0.3:
            base.board = dense_boards_[int(rng.integers(0, 64))].copy()
            obs = env.observation(base._get_obs())
            info = {"board": ref.FeatureVectorObservation(base).observation(base._get_obs())}
        if prefull:
            for row in rng.choice(np.arange(height // 2, height), size=int(rng.integers(1, 4)), replace=False):
                base.board[row, 4:4 + width] = rng.integers(2, 9, width)
            obs = env.observation(base._get_obs())
            info = {"board": ref.FeatureVectorObservation(base).observation(base._get_obs())}
            pboards[k] = plain.observation(base._get_obs())
        boards[k] = base.board
        pieces[k] = base.active_tetromino.id - 2
        mats[k] = matrix16(base.active_tetromino)
        feats[k] = obs
        masks[k] = env.legal_actions_mask
        cur_feats[k] = info["board"]
        k += 1
        legal = np.flatnonzero(env.legal_actions_mask == 1)
        obs, r, d, _, info = env.step(int(rng.choice(legal)))
        if d:
            obs, info = env.reset()
    extra = {"pboards": pboards} if prefull else {}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), width=width, height=height, queue_size=queue_size,
                        boards=boards, pieces=pieces, mats=mats, feats=feats, masks=masks, cur_feats=cur_feats, **extra)
    print(f"{name}: {n_states} states, feature tensors {feats.shape}, max feature {feats.max()}")


def gen_rgb(ref, name, width, height, queue_size, B, T, seed):
    """RgbObservation (board | queue on top, holder at the bottom, colour LUT) along random-policy traces.  The
    wrapper is driven exactly as a user would: env = RgbObservation(Tetris(...)); frames of reset() and every step()."""
    rng = np.random.default_rng(seed)
    # uniform over the 8 actions with extra weight on swap (6) so that the holder strip changes often
    p = np.array([1, 1, 1, 1, 1, 1, 2.5, 0.5]); p /= p.sum()
    actions = rng.choice(8, size=(T, B), p=p).astype(np.int8)
    Hp, Wp = height + 4, width + 8 + 4 * max(1, queue_size)
    frames = np.zeros((T, B, Hp, Wp, 3), np.uint8)
    reset_frames = np.zeros((T, B, Hp, Wp, 3), np.uint8)
    first = np.zeros((B, Hp, Wp, 3), np.uint8)
    done = np.zeros((T, B), np.uint8)
    for i in range(B):
        base = make_env(ref, width, height, True, queue_size)
        env = ref.RgbObservation(base)
        obs, _ = env.reset(seed=1 + i)
        assert obs.shape == (Hp, Wp, 3) and obs.dtype == np.uint8, (obs.shape, obs.dtype)
        first[i] = obs
        for t in range(T):
            obs, r, d, _, info = env.step(int(actions[t, i]))
            frames[t, i], done[t, i] = obs, int(d)
            if d:
                obs, _ = env.reset()
                reset_frames[t, i] = obs
    np.savez_compressed(os.path.join(OUT, name + ".npz"), width=width, height=height, queue_size=queue_size, B=B, T=T,
                        actions=actions, frames=frames, reset_frames=reset_frames, first=first, done=done)
    print(f"{name}: {B*T} frames, episodes={int(done.sum())}, swaps={(actions == 6).sum()}")


def gen_poked(ref, name, width, height, queue_size, n_states, seed, gravity=True, T=16, n_grouped=None):
    """Hand-made states: the state class the reference's own tests use — `env.unwrapped.board[...] = 2`,
    `.active_tetromino = ...`, `.x = ...` (tests/test_base_env/**) — and the one where round 5's two late bugs lived.  Per state:
    reset(seed), then everything a caller can assign is assigned: the playfield (random fill with values 2..8, rows that are ALREADY full,
    fills that reach the spawn rows, and — flagged `bedrock_row` — a playfield row of bedrock cells, which clear_filled_rows never takes,
    envs/tetris.py:493),  the piece (any id / rotation / position the reference's slicing tolerates: x in [0, Wp - n], y in [0, Hp - n],
    incl. on top of filled cells and inside the bedrock; narrowed to x in
    [1, Wp - n - 1], y in [0, Hp - n - 1] because a move probes one cell further and the reference raises there), holder, queue, bag + bag index, has_swapped, game_over.  The generator keeps
    the state reset(seed) left.  Then for EACH of the 8 actions a deep copy of the env takes that action followed by T - 1 random ones
    (no harness reset: the recording of a copy ends with its first done) -> reward / done / lines / observation CRC / x / y / piece per
    transition, full snapshots behind the first transition and at the end.  On the same states: the grouped wrapper's legal mask and
    4W placement boards, EVERY one of its 4W actions stepped (reward, done, lines, board, piece and position after),
    FeatureVectorObservation of the state and of every placement, the RgbObservation frame."""
    import copy

    from tetris_gymnasium.components.tetromino import Tetromino

    rng = np.random.default_rng(seed)
    Hp, Wp, P = height + 4, width + 8, 4
    NA = 4 * width
    ng = n_states if n_grouped is None else n_grouped
    keys = ("x", "y", "piece", "matrix", "queue", "holder_piece", "holder_matrix", "bag", "bag_index", "has_swapped", "game_over")
    st_board = np.zeros((n_states, Hp, Wp), np.uint8)
    st = {k: [None] * n_states for k in keys}
    st_rot = np.zeros(n_states, np.uint8)
    st_holder_rot = np.zeros(n_states, np.uint8)
    st_crc = np.zeros(n_states, np.uint32)
    bedrock_row = np.zeros(n_states, np.uint8)
    seeds = np.zeros(n_states, np.uint64)
    B = 8 * n_states
    actions = np.zeros((T, B), np.int8)
    reward = np.zeros((T, B), np.int16)
    done = np.zeros((T, B), np.uint8)
    lines = np.zeros((T, B), np.uint8)
    valid = np.zeros((T, B), np.uint8)
    crc = np.zeros((T, B), np.uint32)
    xs = np.zeros((T, B), np.uint8); ys = np.zeros((T, B), np.uint8); pieces = np.zeros((T, B), np.uint8)
    first_board = np.zeros((B, Hp, Wp), np.uint8)
    first = {k: [None] * B for k in keys}
    last_board = np.zeros((B, Hp, Wp), np.uint8)
    last = {k: [None] * B for k in keys}
    g_mask = np.zeros((ng, NA), np.uint8)
    g_boards = np.zeros((ng, NA, Hp, Wp), np.uint8)
    # ... and EVERY grouped action stepped from the state (wrappers/grouped.py:209-269; terminate_on_illegal_action=True)
    gs_reward = np.zeros((ng, NA), np.float64); gs_done = np.zeros((ng, NA), np.uint8); gs_lines = np.zeros((ng, NA), np.uint8)
    gs_board = np.zeros((ng, NA, Hp, Wp), np.uint8)
    gs_x = np.zeros((ng, NA), np.uint8); gs_y = np.zeros((ng, NA), np.uint8); gs_piece = np.zeros((ng, NA), np.uint8)
    g_feats = np.zeros((ng, NA, width + 3), np.uint8)
    cur_feats = np.zeros((n_states, width + 3), np.uint8)
    Wrgb = Wp + 4 * max(1, queue_size)
    rgb = np.zeros((n_states, Hp, Wrgb, 3), np.uint8)
    t0 = time.time()

    def rotated(env, pid, k):
        t = copy.deepcopy(env.tetrominoes[pid])
        for _ in range(k):
            t = env.rotate(t, True)
        return t

    for i in range(n_states):
        env = make_env(ref, width, height, gravity, queue_size)
        seeds[i] = 1000 + i
        env.reset(seed=int(seeds[i]))
        # ---- playfield ----
        mode = i % 6
        pf = np.zeros((height, width), np.uint8)
        if mode >= 1:
            top = int(rng.integers(0, height - 2)) if mode == 4 else int(rng.integers(height // 3, height - 1))
            dens = float(rng.uniform(0.35, 0.9))
            fill = rng.random((height - top, width)) < dens
            pf[top:] = np.where(fill, rng.integers(2, 9, (height - top, width)), 0).astype(np.uint8)
        if mode in (2, 3, 5):  # rows that are already full (a board only a caller can provide)
            for row in rng.choice(np.arange(height // 4, height), size=int(rng.integers(1, 4)), replace=False):
                pf[row] = rng.integers(2, 9, width)
        if mode == 5 and i % 12 == 5:  # a playfield row of bedrock cells
            row = int(rng.integers(height // 2, height))
            pf[row] = 1
            bedrock_row[i] = 1
        env.board[:height, P:P + width] = pf
        # ---- piece ----
        for attempt in range(64):
            pid, k = int(rng.integers(0, 7)), int(rng.integers(0, 4))
            t = rotated(env, pid, k)
            n = t.matrix.shape[0]
            # (every position a step may probe — x - 1, x + 1, y + 1 — must slice inside the padded board: beyond it the reference raises)
            x, y = int(rng.integers(1, Wp - n)), int(rng.integers(0, Hp - n))
            if i % 3 == 0 or not env.collision(t, x, y):  # (a third of the states: wherever it falls, incl. on filled cells / bedrock)
                break
        if i % 16 == 9:  # the deepest row the slicing tolerates: an O piece wholly inside the bottom bedrock (then e.g. a hold swap)
            pid, k = 1, int(rng.integers(0, 4))
            t = rotated(env, pid, k)
            x, y = int(rng.integers(1, Wp - 2)), Hp - 2 - 1
        env.active_tetromino, env.x, env.y = t, x, y
        st_rot[i] = k
        # ---- holder / queue / bag / flags ----
        if rng.random() < 0.6:
            hk = int(rng.integers(0, 4))
            env.holder.queue.clear()
            env.holder.queue.append(rotated(env, int(rng.integers(0, 7)), hk))
            st_holder_rot[i] = hk
        env.has_swapped = bool(rng.random() < 0.4) and i % 16 != 9
        env.game_over = bool(rng.random() < 0.1)
        env.queue.queue.clear()
        for _ in range(queue_size):
            env.queue.queue.append(int(rng.integers(0, 7)))
        env.randomizer.bag = rng.permutation(7).astype(np.int8)
        env.randomizer.index = int(rng.integers(0, 7))
        # ---- the state ----
        s0 = snapshot(env)
        st_board[i] = s0["board"]
        for k_ in keys:
            st[k_][i] = s0[k_]
        st_crc[i] = obs_crc(env._get_obs())
        cur_feats[i] = ref.FeatureVectorObservation(env).observation(env._get_obs())
        rgb[i] = ref.RgbObservation(env).observation(env._get_obs())
        if i < ng:
            e2 = copy.deepcopy(env)
            gw = ref.GroupedActionsObservations(e2)
            g_boards[i] = gw.observation(e2._get_obs())
            g_mask[i] = gw.legal_actions_mask
            e3 = copy.deepcopy(env)
            gf = ref.GroupedActionsObservations(e3, observation_wrappers=[ref.FeatureVectorObservation(e3)])
            g_feats[i] = gf.observation(e3._get_obs())
            for a in range(NA):
                e4 = copy.deepcopy(env)
                g4 = ref.GroupedActionsObservations(e4)
                g4.observation(e4._get_obs())  # (the wrapper steps against the mask of its last observation)
                _, r4, d4, _, info4 = g4.step(a)
                gs_reward[i, a], gs_done[i, a], gs_lines[i, a] = float(r4), int(d4), int(info4["lines_cleared"])
                gs_board[i, a] = e4.board
                gs_x[i, a], gs_y[i, a], gs_piece[i, a] = e4.x, e4.y, e4.active_tetromino.id - 2
        # ---- every action, then random ones ----
        for a in range(8):
            j = 8 * i + a
            e = copy.deepcopy(env)
            for t_ in range(T):
                act = a if t_ == 0 else int(rng.integers(0, 8))
                actions[t_, j] = act
                obs, r, d, _, info = e.step(act)
                assert r == int(r)
                valid[t_, j] = 1
                reward[t_, j], done[t_, j], lines[t_, j] = int(r), int(d), int(info["lines_cleared"])
                crc[t_, j] = obs_crc(obs)
                xs[t_, j], ys[t_, j], pieces[t_, j] = e.x, e.y, e.active_tetromino.id - 2
                if t_ == 0:
                    s1 = snapshot(e)
                    first_board[j] = s1["board"]
                    for k_ in keys:
                        first[k_][j] = s1[k_]
                if d:
                    break
            s2 = snapshot(e)
            last_board[j] = s2["board"]
            for k_ in keys:
                last[k_][j] = s2[k_]
    out = dict(width=width, height=height, gravity=int(gravity), queue_size=queue_size, n_states=n_states, T=T, seeds=seeds,
               st_board=st_board, st_rot=st_rot, st_holder_rot=st_holder_rot, st_crc=st_crc, bedrock_row=bedrock_row,
               actions=actions, reward=reward, done=done, lines=lines, valid=valid, crc=crc, x=xs, y=ys, piece=pieces,
               first_board=first_board, last_board=last_board, g_mask=g_mask, g_boards=g_boards, g_feats=g_feats,
               cur_feats=cur_feats, rgb=rgb, gs_reward=gs_reward, gs_done=gs_done, gs_lines=gs_lines, gs_board=gs_board,
               gs_x=gs_x, gs_y=gs_y, gs_piece=gs_piece)
    for k_ in keys:
        out["st_" + k_] = np.array(st[k_]).astype(np.uint8)
        out["first_" + k_] = np.array(first[k_]).astype(np.uint8)
        out["last_" + k_] = np.array(last[k_]).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {n_states} states x 8 actions x <= {T} transitions in {time.time()-t0:.1f}s  valid={int(valid.sum())} "
          f"dones={int(done.sum())} first-step locks={(reward[0] != 0).sum()} lines={int(lines.sum())} (first step {int(lines[0].sum())}) "
          f"bedrock-row states={int(bedrock_row.sum())} game_over states={int(np.array(st['game_over']).sum())}")


def gen_grouped_rgb(ref, name, width, height, queue_size, T, seed):
    """GroupedActionsObservations(env, observation_wrappers=[RgbObservation(env)]) (wrappers/grouped.py:181-204 feeding
    wrappers/observation.py:38-74): one RGB image per placement — the placement board (ones for an illegal placement, zeros for a
    game-over one) beside the env's holder / queue strips — plus info["board"], the image of the env's own observation."""
    rng = np.random.default_rng(seed)
    base = make_env(ref, width, height, True, queue_size)
    env = ref.GroupedActionsObservations(base, observation_wrappers=[ref.RgbObservation(base)])
    obs, info = env.reset(seed=seed)
    NA = 4 * width
    shape = obs.shape
    assert shape[0] == NA and shape[-1] == 3, shape
    first_obs, first_board, first_mask = obs.astype(np.uint8), info["board"].astype(np.uint8), env.legal_actions_mask.astype(np.uint8)
    actions = np.zeros(T, np.int16)
    frames = np.zeros((T,) + shape, np.uint8)
    boards = np.zeros((T,) + shape[1:], np.uint8)
    masks = np.zeros((T, NA), np.uint8)
    done = np.zeros(T, np.uint8)
    reward = np.zeros(T, np.float64)
    for t in range(T):
        legal = np.flatnonzero(env.legal_actions_mask == 1)
        a = int(rng.choice(legal)) if len(legal) else 0
        if t % 5 == 4:  # a hold in between, so that the holder strip fills (the base env's own swap action)
            base.step(base.actions.swap)
            env.observation(base._get_obs())
            legal = np.flatnonzero(env.legal_actions_mask == 1)
            a = int(rng.choice(legal)) if len(legal) else 0
        actions[t] = a
        obs, r, d, _, info = env.step(a)
        frames[t], boards[t], masks[t], done[t], reward[t] = obs, info["board"], env.legal_actions_mask, int(d), float(r)
        if d:
            obs, info = env.reset()
            frames[t], boards[t], masks[t] = obs, info["board"], env.legal_actions_mask  # (what the next action is drawn from)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), width=width, height=height, queue_size=queue_size, T=T, seed=seed,
                        actions=actions, frames=frames, boards=boards, masks=masks, done=done, reward=reward,
                        first_obs=first_obs, first_board=first_board, first_mask=first_mask)
    print(f"{name}: {T} grouped steps with RGB placements {frames.shape}, episodes={int(done.sum())}")


def gen_ref_tests_grouped(ref):
    """Data held by the reference's own grouped tests: the mock board (tests/helpers/mock.py:5-47 applied to
    a fresh env), the CSV board after action 21 (tests/test_grouped_env/expected_result_i_placement.csv) and
    the 40-entry legal mask the reference computes on the mock board with a vertical I piece
    (tests/test_grouped_env/actions/test_grouped_actions.py:11-31)."""
    env = ref.Tetris()
    g = ref.GroupedActionsObservations(env)
    g.reset(seed=42)
    board = np.copy(env.board)
    th = env.height // 2
    board[th:env.height, env.padding:-(env.padding + 1)] = 2
    board[th - 1, env.padding + 1] = 2
    board[th - 1, env.padding + 4] = 2
    board[th - 1, env.padding + 5] = 2
    board[th + 2, env.padding + 2] = 0
    board[th + 4, env.padding + 3] = 0
    board[th + 6, env.padding + 6] = 0
    import copy

    vi = copy.deepcopy(ref.Tetris.TETROMINOES[0])
    vi.matrix = np.rot90(vi.matrix)
    env.board = board.copy()
    env.active_tetromino = vi
    obs = {"board": board, "holder": None, "queue": None, "active_tetromino_mask": np.zeros_like(board)}
    boards40 = g.observation(obs).astype(np.uint8)
    mask = g.legal_actions_mask.astype(np.uint8).copy()
    g.step(5 * 4 + 1)
    after = env.board.copy()
    csv = np.genfromtxt("/root/reference/tests/test_grouped_env/expected_result_i_placement.csv",
                        delimiter=",").astype(np.uint8)
    assert np.array_equal(after, csv)
    expected_mask = np.array(
        [[0, 1, 1, 1, 1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1, 1, 1, 1, 0], [1] * 10, [0, 0, 1, 1, 1, 1, 1, 1, 1, 0]],
        np.uint8).reshape(40, order="F")
    assert np.array_equal(mask, expected_mask)
    # feature-vector expectations of tests/helpers/mock.py:35-39
    np.savez_compressed(os.path.join(OUT, "ref_tests_grouped.npz"), mock_board=board, csv_board=csv, mask=mask,
                        boards40=boards40, heights=np.array([10, 11, 10, 10, 11, 11, 10, 10, 10, 0], np.uint8),
                        max_height=11, holes=3, bumpiness=14)
    print("ref_tests_grouped: ok (CSV and mask reproduce)")


def gen_features(ref):
    """FeatureVectorObservation on random mid-game observations (wrappers/observation.py:236-278)."""
    rng = np.random.default_rng(99)
    N = 256
    boards = np.zeros((N, 24, 18), np.uint8)
    masks = np.zeros((N, 24, 18), np.uint8)
    feats = np.zeros((N, 13), np.uint8)
    env = ref.Tetris()
    fv = ref.FeatureVectorObservation(env)
    env.reset(seed=7)
    k = 0
    while k < N:
        obs, r, d, _, _ = env.step(int(rng.integers(0, 8)))
        if d:
            env.reset()
            continue
        if rng.random() < 0.5:
            boards[k] = obs["board"]
            masks[k] = obs["active_tetromino_mask"]
            feats[k] = fv.observation({k2: v.copy() for k2, v in obs.items()})
            k += 1
    np.savez_compressed(os.path.join(OUT, "features.npz"), boards=boards, masks=masks, feats=feats)
    print("features: ok", feats[:3])


# ActionsMapping field values in declaration order: move_left, move_right, move_down, rotate_clockwise,
# rotate_counterclockwise, hard_drop, swap, no_op
MAP_PERM = (3, 7, 0, 5, 1, 2, 4, 6)
MAP_DUP = (0, 1, 2, 3, 3, 5, 6, 12)  # ccw duplicates cw (unreachable), ids 4 and 7 belong to no field
# hard_drop sharing its value with a field the elif chain tests EARLIER (envs/tetris.py:222-259): that id runs the other operation and
# then skips gravity (`action != self.actions.hard_drop` is false); no id hard-drops, pieces lock by gravity alone
MAP_ALIAS_LEFT = (5, 1, 2, 3, 4, 5, 6, 7)  # move_left == hard_drop == 5; id 0 belongs to no field
MAP_ALIAS_SWAP = (0, 1, 2, 6, 4, 6, 6, 7)  # rotate_clockwise == hard_drop == swap == 6: the rotation wins, no gravity; ids 3, 5: no field


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    jobs = {
        "pcg64": gen_pcg64_streams,
        "trace_10x20_g1": lambda: run_trace(ref, "trace_10x20_g1", 10, 20, True, 4, 96, 512, 12345),
        "trace_10x20_g0": lambda: run_trace(ref, "trace_10x20_g0", 10, 20, False, 4, 48, 512, 12346),
        "trace_20x40_g1": lambda: run_trace(ref, "trace_20x40_g1", 20, 40, True, 7, 32, 768, 12347),
        "trace_6x12_g1": lambda: run_trace(ref, "trace_6x12_g1", 6, 12, True, 4, 32, 256, 12348),
        # sizes the engine compiles at run time (hiprtc), one with 98-byte slabs (7x14: env slabs at 2-byte alignment) and a 2-deep
        # queue, one larger board with the longest queue and gravity off
        "trace_7x14_q2_g1": lambda: run_trace(ref, "trace_7x14_q2_g1", 7, 14, True, 2, 40, 320, 12349),
        "trace_12x24_q7_g0": lambda: run_trace(ref, "trace_12x24_q7_g0", 12, 24, False, 7, 24, 320, 12350),
        "clears_10x20": lambda: gen_clears(ref, "clears_10x20", 10, 20, 4, 96, 256, 777, 5),
        "clears_20x40": lambda: gen_clears(ref, "clears_20x40", 20, 40, 7, 32, 384, 778, 6),
        "grouped_10x20": lambda: gen_grouped(ref, "grouped_10x20", 10, 20, 4, 16, 96, 31, 48),
        "grouped_20x40": lambda: gen_grouped(ref, "grouped_20x40", 20, 40, 7, 4, 64, 32, 12),
        "grouped_clears_10x20": lambda: gen_grouped(ref, "grouped_clears_10x20", 10, 20, 4, 16, 96, 33, 16,
                                                    init_boards=dense_boards(10, 20, 16, 8)[0]),
        "grouped_features_10x20": lambda: gen_grouped_features(ref, "grouped_features_10x20", 10, 20, 4, 192, 41, dense=True),
        "grouped_features_20x40": lambda: gen_grouped_features(ref, "grouped_features_20x40", 20, 40, 7, 24, 42, dense=True),
        "grouped_features_prefull_6x12": lambda: gen_grouped_features(ref, "grouped_features_prefull_6x12", 6, 12, 4, 48, 43, prefull=True),
        "grouped_features_prefull_10x20": lambda: gen_grouped_features(ref, "grouped_features_prefull_10x20", 10, 20, 4, 32, 44, prefull=True),
        "ref_tests_grouped": lambda: gen_ref_tests_grouped(ref),
        "features": lambda: gen_features(ref),
        # non-stock ActionsMapping / RewardsMapping (envs/tetris.py:83-84, 189-190): a permutation of the ids with float rewards;
        # line clears with integer rewards; a mapping with a duplicate value (the first field of the elif chain wins,
        # :222-256) and ids that no field holds (gravity only); the grouped wrapper's invalid_action reward (grouped.py:238)
        "mapped_10x20_g1": lambda: run_trace(ref, "mapped_10x20_g1", 10, 20, True, 4, 48, 384, 12360,
                                             action_ids=MAP_PERM, rewards=(0.5, 10, -5.25, -0.25)),
        "mapped_10x20_g0": lambda: run_trace(ref, "mapped_10x20_g0", 10, 20, False, 4, 24, 256, 12361,
                                             action_ids=MAP_DUP, rewards=(2, 1, -7, -0.1)),
        "mapped_alias_left_10x20": lambda: run_trace(ref, "mapped_alias_left_10x20", 10, 20, True, 4, 32, 320, 12362,
                                                     action_ids=MAP_ALIAS_LEFT, rewards=(1, 1, 0, -0.1)),
        "mapped_alias_swap_10x20": lambda: run_trace(ref, "mapped_alias_swap_10x20", 10, 20, True, 4, 32, 320, 12363,
                                                     action_ids=MAP_ALIAS_SWAP, rewards=(0.25, 1, -2, -0.1)),
        "mapped_clears_10x20": lambda: gen_clears(ref, "mapped_clears_10x20", 10, 20, 4, 48, 256, 779, 7,
                                                  action_ids=MAP_PERM, rewards=(0, 10, -5, -1)),
        "mapped_grouped_10x20": lambda: gen_grouped(ref, "mapped_grouped_10x20", 10, 20, 4, 16, 96, 34, 0,
                                                    action_ids=MAP_PERM, rewards=(3, 1, -4, -0.375),
                                                    terminate_on_illegal=False, p_any=0.25),
        "rgb_10x20": lambda: gen_rgb(ref, "rgb_10x20", 10, 20, 4, 8, 160, 51),
        "rgb_20x40": lambda: gen_rgb(ref, "rgb_20x40", 20, 40, 7, 2, 200, 52),
        "grouped_rgb_10x20": lambda: gen_grouped_rgb(ref, "grouped_rgb_10x20", 10, 20, 4, 40, 71),
        "poked_10x20": lambda: gen_poked(ref, "poked_10x20", 10, 20, 4, 192, 61),
        "poked_10x20_g0": lambda: gen_poked(ref, "poked_10x20_g0", 10, 20, 4, 72, 62, gravity=False, n_grouped=0),
        "poked_20x40": lambda: gen_poked(ref, "poked_20x40", 20, 40, 7, 60, 63, n_grouped=12),
    }
    for name, fn in jobs.items():
        if args.only and args.only != name:
            continue
        fn()


if __name__ == "__main__":
    sys.exit(main())
