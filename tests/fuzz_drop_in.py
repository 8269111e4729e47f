#!/usr/bin/env python3
"""Runs ON THE GPU BOX: randomised differential test of the single-env drop-in `Tetris` and its wrapper classes against the C oracle
(test infrastructure).  One base env, three wrappers around it (grouped actions, RGB observation, feature vectors) and a random walk
over everything a caller can do in any order: base steps, grouped steps through the wrapper, resets (seeded / unseeded), state
snapshots and restores, attribute writes, in-place edits of handed-out arrays, plain attribute reads, wrapper observations.  After
every operation the env's observation — and at random the wrappers' observations, the action mask and the host-side attributes — must
equal the oracle's.  (The wrappers keep an auxiliary engine / write state back on demand: this is the test of that bookkeeping.)
usage: python tests/fuzz_drop_in.py [n_walks] [seed]     (long runs; tests/test_gpu_fuzz.py runs a bounded slice of run_walk under pytest -m gpu)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tetris_gymnasium_amd.envs import Tetris  # noqa: E402
from tetris_gymnasium_amd.wrappers import FeatureVectorObservation, GroupedActionsObservations, RgbObservation  # noqa: E402

OPS = int(os.environ.get("FUZZ_OPS", 250))
SIZES = [(10, 20), (6, 12), (8, 16), (10, 24), (12, 24)]


def run_walk(SEED, wi, OPS=OPS):
    """Walk `wi` of seed `SEED`: raises AssertionError naming the configuration and the last operations, returns the line the CLI prints."""
    r = np.random.default_rng([SEED, wi])
    W, H = SIZES[int(r.integers(len(SIZES)))]
    Q = int(r.integers(1, 8))
    gravity = bool(r.random() < 0.8)
    with_fv = bool(r.random() < 0.5)
    toi = bool(r.random() < 0.5)
    base = Tetris(width=W, height=H, gravity=gravity, queue_size=Q)
    grp = GroupedActionsObservations(base, observation_wrappers=[FeatureVectorObservation(base)] if with_fv else None,
                                     terminate_on_illegal_action=toi)
    rgb, fv = RgbObservation(base), FeatureVectorObservation(base)
    o = po.OracleBatch(1, W, H, po.RULESET_NUMPY, gravity, Q)
    tag = f"{W}x{H} Q={Q} g={int(gravity)} fv={int(with_fv)} toi={int(toi)}"
    s0 = int(r.integers(1, 1 << 30))
    obs, _ = base.reset(seed=s0)
    o.reset(np.array([s0], np.uint64))
    saved, log = [], []

    def check(what):
        ref = o.observe(po.OBS_DICT)
        got = base._get_obs()
        for k in ref:
            assert np.array_equal(got[k], ref[k][0]), f"base observation[{k}] differs after {what} [{tag}] ops: {log[-8:]}"
        u = r.random()
        if u < 0.25:
            assert np.array_equal(rgb.observation(), o.observe(po.OBS_RGB)[0]), f"rgb differs after {what} [{tag}] ops: {log[-8:]}"
        elif u < 0.5:
            assert np.array_equal(fv.observation(), o.features()[0]), f"features differ after {what} [{tag}] ops: {log[-8:]}"
        elif u < 0.7:
            m, b = o.grouped_enumerate(boards=True)
            g = grp.observation()
            assert np.array_equal(g, o.grouped_features()[0] if with_fv else b[0]), f"grouped observation differs after {what} [{tag}] ops: {log[-8:]}"
            assert np.array_equal(grp.legal_actions_mask, m[0]), f"action mask differs after {what} [{tag}] ops: {log[-8:]}"
        elif u < 0.85:
            bo, ro, _ = o.get_state()
            assert np.array_equal(base.board[:H, 4:4 + W], bo[0]) and (base.x, base.y) == (int(ro["x"][0]), int(ro["y"][0])), \
                f"host attributes differ after {what} [{tag}] ops: {log[-8:]}"

    check("reset")
    for k in range(OPS):
        u = r.random()
        game_over = bool(o.get_state()[1]["flags"][0] & 1)
        if u < 0.35:  # a base step
            a = int(r.integers(0, 8))
            log.append(f"step({a})")
            ob, rew, term, _, info = base.step(a)
            ro, to, lo = o.step(np.array([a]))
            assert rew == o.last_rewards()[0] and term == bool(to[0]) and info["lines_cleared"] == lo[0], f"step outputs differ [{tag}] ops: {log[-8:]}"
            if term and r.random() < 0.7:
                log.append("reset()")
                base.reset(); o.reset()
        elif u < 0.65:  # a grouped step through the wrapper
            m = o.grouped_enumerate()[0]
            grp.observation()  # (the wrapper's mask is the one of its last observation, as in the reference)
            legal = np.flatnonzero(m)
            a = int(r.choice(legal)) if (len(legal) and r.random() < 0.9) else int(r.integers(0, 4 * W))
            log.append(f"grouped({a})")
            g, rew, term, _, info = grp.step(a)
            ro, to, lo, ill = o.grouped_step(np.array([a]), terminate_on_illegal=toi)
            want = -0.1 if ill[0] else o.last_rewards()[0]
            assert rew == want and term == bool(to[0]), f"grouped step outputs differ: {rew} {term} vs {want} {to[0]} [{tag}] ops: {log[-8:]}"
            if ill[0] and toi:  # (the reference returns before touching the env: nothing to compare but the constant observation)
                assert np.all(g == grp.observation_space.high)
            if term and r.random() < 0.8:
                log.append("grouped reset()")
                grp.reset(); o.reset()
        elif u < 0.72:
            sd = int(r.integers(1, 1 << 30)) if r.random() < 0.5 else None
            log.append(f"reset({sd})")
            (grp if r.random() < 0.5 else base).reset(seed=sd)
            o.reset(None if sd is None else np.array([sd], np.uint64))
        elif u < 0.80:
            log.append("snapshot")
            saved.append((base.get_state(), o.get_state()))
        elif u < 0.86 and saved:
            log.append("restore")
            sb, so = saved[int(r.integers(len(saved)))]
            base.set_state(sb); o.set_state(*so)
        elif u < 0.92 and not game_over:
            if not base.collision(base.active_tetromino, base.x, base.y + 1):
                log.append("y += 1")
                base.y = base.y + 1
                b, rc, g_ = o.get_state(); rc["y"] += 1; o.set_state(b, rc, g_)
        elif u < 0.97:
            row, col = int(r.integers(H // 2, H)), int(r.integers(0, W))
            board = base.board  # handed out ...
            if board[row, 4 + col] == 0 and not base.active_tetromino is None:
                log.append(f"board[{row},{col}] = 5 in place")
                board[row, 4 + col] = 5  # ... and edited in place
                b, rc, g_ = o.get_state(); b[0, row, col] = 5; o.set_state(b, rc, g_)
                # (a cell under the active piece would make the next projection a collision in both worlds alike)
        else:
            log.append("read attributes")
            _ = (base.board, base.queue.get_queue(), base.holder.get_tetrominoes(), base.active_tetromino, base.has_swapped, base.game_over)
        check(log[-1] if log else "nothing")
    grp.close()
    return f"ok   {tag}: {OPS} operations"


if __name__ == "__main__":
    NW = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    for wi in range(NW):
        print(run_walk(SEED, wi), flush=True)
    print(f"{NW} walks, seed {SEED}: {time.time() - t0:.0f} s")
