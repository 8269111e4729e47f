#!/usr/bin/env python3
"""Runs ON THE GPU BOX: randomised differential test of the `TetrisVectorEnv` adapter against the C oracle (test infrastructure):
random batch size, shard count (several handles on one device), autoreset mode, observation surface, rule-set, queue length,
grouped action set, mappings, copy / no-copy, and per step a random choice between the host path (NumPy results, one call into the
C ABI) and the device-resident path (views downloaded for the comparison).  Observations, rewards (as Python computes them),
terminations, `infos["lines_cleared"]`, the episode statistics and the grouped action mask must equal the oracle's at every step.
usage: python tests/fuzz_vector_env.py [n_configs] [seed]     (long runs; tests/test_gpu_fuzz.py runs a bounded slice of run_config under pytest -m gpu)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tetris_gymnasium_amd.vector import TetrisVectorEnv  # noqa: E402

STEPS = int(os.environ.get("FUZZ_STEPS", 160))
SIZES = [(10, 20), (6, 12), (8, 16), (10, 24), (12, 24), (20, 40)]
PO_OBS = {"dict": po.OBS_DICT, "board": po.OBS_BOARD, "rgb": po.OBS_RGB, "functional": po.OBS_FN}


def same_obs(got, ref, where):
    if isinstance(ref, dict):
        for k in ref:
            assert np.array_equal(np.asarray(got[k]), ref[k]), f"observation[{k}] differs {where}"
    else:
        assert np.array_equal(np.asarray(got), ref), f"observation differs {where}"


def run_config(SEED, ci, STEPS=STEPS):
    """Configuration `ci` of seed `SEED`: raises AssertionError naming the configuration, returns the line the CLI prints."""
    r = np.random.default_rng([SEED, ci])
    W, H = SIZES[int(r.integers(len(SIZES)))]
    functional = r.random() < 0.25
    grouped = (not functional) and r.random() < 0.25
    kind = "functional" if functional else ("board" if grouped else ["dict", "board", "rgb"][int(r.integers(3))])
    Q = int(r.integers(1, 8)) if kind != "rgb" else 4  # (any length works for RGB; 4 keeps the in-kernel pack in the draw most of the time)
    if kind == "rgb" and r.random() < 0.3:
        Q = int(r.integers(1, 8))
    mode = ["next_step", "same_step"][int(r.integers(2))]
    B = int(r.choice([5, 64, 130, 257]))
    shards = int(r.integers(1, 5))
    copy = bool(r.random() < 0.6)
    toi = bool(r.random() < 0.5)
    ids = rws = None
    if not functional and not grouped and r.random() < 0.3:
        ids = tuple(int(v) for v in r.permutation(8))
        rws = (float(r.integers(-4, 5)) / 2, 1.0, float(r.integers(-9, 1)), -0.25)
    tag = (f"{W}x{H} {kind} Q={Q} {mode} B={B} shards={shards} copy={int(copy)} grouped={int(grouped)}{'/toi' if grouped and toi else ''}"
           f"{' mapped' if ids else ''}")
    env = TetrisVectorEnv(B, W, H, ruleset="functional" if functional else "numpy", observation=kind, queue_size=Q, autoreset_mode=mode,
                          devices=(0,) * shards, grouped=grouped, terminate_on_illegal_action=toi, copy=copy,
                          actions_mapping=ids, rewards_mapping=rws)
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL if functional else po.RULESET_NUMPY, True, Q, action_ids=ids, rewards=rws)
    s0 = int(r.integers(1, 1 << 30))
    obs, info = env.reset(seed=s0)
    o.reset(np.arange(s0, s0 + B, dtype=np.uint64))
    same_obs(obs, o.observe(PO_OBS[kind]), f"after reset [{tag}]")
    na = 4 * W if grouped else (7 if functional else 8)
    ar = po.AUTORESET_NEXT_STEP if mode == "next_step" else po.AUTORESET_SAME_STEP
    ret, ln, prev, n_eps = np.zeros(B), np.zeros(B, int), np.zeros(B, bool), 0
    for t in range(STEPS):
        a = r.integers(0, na, B)
        if grouped and r.random() < 0.8:  # mostly legal placements, so that episodes last
            m = o.grouped_enumerate()
            a = np.array([int(r.choice(np.flatnonzero(row))) if row.any() else 0 for row in m])
        where = f"at step {t} [{tag}]"
        host = r.random() < 0.7
        if host:
            obs, rew, term, trunc, infos = env.step(a)
        else:
            res = env.step(a, to_host=False)
            parts = res if isinstance(res, list) else [res]
            outs = [{k: np.asarray(v) for k, v in p[1].items()} for p in parts]
            term = np.concatenate([x["terminated"] for x in outs]).astype(bool)
            lines_dev = np.concatenate([x["lines"] for x in outs])
            ob = [p[0] for p in parts]
            if isinstance(ob[0], dict):
                obs = {k: np.concatenate([np.asarray(x[k]) for x in ob]) for k in ob[0]}
            else:
                obs = np.concatenate([np.asarray(x) for x in ob])
        if grouped:
            ro, to, lo, _ = o.grouped_step(a, terminate_on_illegal=toi, autoreset=ar)
        else:
            ro, to, lo = o.step(a, autoreset=ar)
        assert np.array_equal(term, to.astype(bool)), f"terminations differ {where}"
        same_obs(obs, o.observe(PO_OBS[kind]), where)
        if host:
            want = o.last_rewards() if not functional else ro.astype(np.float64)
            assert rew.dtype == np.float64 and np.array_equal(rew, want), f"rewards differ {where}: {rew[:6]} vs {want[:6]}"
            assert np.array_equal(infos["lines_cleared"], lo) and not trunc.any(), f"lines differ {where}"
            if grouped:
                assert np.array_equal(infos["action_mask"], o.grouped_enumerate()), f"action mask differs {where}"
            count = ~prev if mode == "next_step" else np.ones(B, bool)
            ret[count] += rew[count]; ln[count] += 1
            if term.any():
                assert np.array_equal(infos["_episode"], term), f"_episode differs {where}"
                assert np.allclose(infos["episode"]["r"][term], ret[term]) and np.array_equal(infos["episode"]["l"][term], ln[term]), f"episode statistics differ {where}"
        else:
            assert np.array_equal(lines_dev, lo), f"lines differ {where}"
            count = ~prev if mode == "next_step" else np.ones(B, bool)
            rw = o.last_rewards() if not functional else ro.astype(np.float64)
            ret[count] += rw[count]; ln[count] += 1
        prev = term.copy()
        n_eps += int(term.sum()); ret[term] = 0; ln[term] = 0
    env.close()
    return f"ok   {tag}: {STEPS} steps, {n_eps} episodes ended"


if __name__ == "__main__":
    NCFG = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    for ci in range(NCFG):
        print(run_config(SEED, ci), flush=True)
    print(f"{NCFG} configurations, seed {SEED}: {time.time() - t0:.0f} s")
