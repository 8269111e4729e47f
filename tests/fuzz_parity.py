#!/usr/bin/env python3
"""Runs ON THE GPU BOX: randomised differential test of the engine against the C oracle (test infrastructure) — where the pytest
suite walks chosen configurations, this draws them: board size (built in and compiled at run time), rule-set, queue / bag length,
gravity, autoreset mode, observation surface, action set, launch knobs (transitions per launch, ring depth, RNG-role cadence),
mappings, batch size; per configuration a few hundred transitions in rollouts of random length with, in between, masked and
seeded resets and a get_state / set_state round trip.  Every transition's outputs and observation and, after every rollout, the
whole state (boards, records, generators) must equal the oracle's.
usage: python tests/fuzz_parity.py [n_configs] [seed]     (long runs; tests/test_gpu_fuzz.py runs a bounded slice of run_config under pytest -m gpu)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402

MAX_RTC = int(os.environ.get("FUZZ_RTC_SIZES", 3))  # board sizes compiled at run time (tens of seconds each on a fresh box)
BUILT_IN = [(10, 20), (20, 40), (6, 12), (8, 16), (10, 24), (12, 24)]
PO_OBS = {N.OBS_BOARD: po.OBS_BOARD, N.OBS_FN: po.OBS_FN, N.OBS_DICT: po.OBS_DICT, N.OBS_RGB: po.OBS_RGB}
PO_AR = {N.AUTORESET_NONE: po.AUTORESET_NONE, N.AUTORESET_SAME_STEP: po.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP: po.AUTORESET_NEXT_STEP}


def draw_rtc_sizes(seed, n=MAX_RTC):
    top = np.random.default_rng(seed)
    return [(int(top.integers(4, 17)), int(top.integers(6, 31))) for _ in range(n)]


def same_state(e, o, numpy_rules, where):
    be, re_, ge = e.get_state()
    bo, ro, go = o.get_state()
    assert np.array_equal(be, bo), f"board differs {where}"
    for f in re_.dtype.names:
        if f != "reserved":
            assert np.array_equal(re_[f], ro[f]), f"rec.{f} differs {where}"
    if numpy_rules:
        for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
            assert np.array_equal(ge[f], go[f]), f"rng.{f} differs {where}"
    else:
        assert np.array_equal(ge["key"], go["key"]), f"rng.key differs {where}"


def same_obs(got, ref, where):
    if isinstance(ref, dict):
        for k in ref:
            assert np.array_equal(got[k], ref[k]), f"observation[{k}] differs {where}"
    else:
        assert np.array_equal(got, ref), f"observation differs {where}"


def run_config(SEED, ci, rtc_sizes=(), steps=None):
    """Configuration `ci` of seed `SEED` (the draw depends on nothing else but the run-time-compiled sizes offered): raises AssertionError
    naming the configuration, returns the line the CLI prints."""
    r = np.random.default_rng([SEED, ci])
    W, H = BUILT_IN[int(r.integers(len(BUILT_IN)))] if (r.random() < 0.75 or not rtc_sizes) else rtc_sizes[int(r.integers(len(rtc_sizes)))]
    numpy_rules = bool(r.random() < 0.65)
    rs = N.RULESET_NUMPY if numpy_rules else N.RULESET_FUNCTIONAL
    Q = int(r.integers(1, 8))
    gravity = bool(r.random() < 0.8)
    autoreset = [N.AUTORESET_NONE, N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP][int(r.integers(3))]
    grouped = numpy_rules and r.random() < 0.25
    toi = bool(r.random() < 0.5)
    obs = (N.OBS_BOARD if grouped else [N.OBS_BOARD, N.OBS_DICT, N.OBS_RGB, N.OBS_NONE][int(r.integers(4))]) if numpy_rules else N.OBS_FN
    B = int(r.choice([1, 3, 17, 64, 100, 129, 200, 333, 2048 if W * H <= 240 else 260]))
    streams = 1
    if r.random() < 0.06 and W * H <= 200 and not grouped and obs in (N.OBS_BOARD, N.OBS_FN, N.OBS_NONE):
        B, streams = 16384 + 128 * int(r.integers(0, 3)), 2  # env shards on two streams (tg_config.streams)
    packed = r.random() < 0.2  # TG_PACKED=1: the board as nibbles in HBM where the geometry has that build (else ignored)
    if grouped:
        B = (B + 3) // 4 * 4  # (tg_rollout: int8 action rows of a grouped handle are packed by fours)
    fuse, ring_fill, rng_every = int(r.choice([1, 2, 5, 16])), int(r.integers(0, 4)), int(r.integers(1, 4))
    ids = rws = None
    if numpy_rules and not grouped and r.random() < 0.3:
        ids = tuple(int(v) for v in r.permutation(8))
        rws = (float(r.integers(-4, 5)) / 2, 1.0, float(r.integers(-9, 1)), -0.25)
    tag = (f"{W}x{H} {'numpy' if numpy_rules else 'functional'} Q={Q} g={int(gravity)} autoreset={autoreset} obs={obs} grouped={int(grouped)}"
           f"{'/toi' if grouped and toi else ''} B={B}{' streams=2' if streams > 1 else ''}{' packed' if packed else ''} fuse={fuse} ring={ring_fill} rng_every={rng_every}{' mapped' if ids else ''}")
    packed_before = os.environ.get("TG_PACKED")
    os.environ["TG_PACKED"] = "1" if packed else "0"
    try:
        e = Engine(B, W, H, ruleset=rs, gravity=gravity, queue_size=Q, obs_mode=obs, autoreset=autoreset, grouped=grouped,
                   terminate_on_illegal=toi, fuse=fuse, ring_fill=ring_fill, rng_every=rng_every, actions_mapping=ids, rewards_mapping=rws, streams=streams)
    except Exception as ex:  # (a configuration tg_create refuses is not a parity failure: say so and go on)
        return f"skip {tag}: {str(ex)[:120]}"
    finally:  # (read by tg_create only: the caller's environment is left as it was)
        if packed_before is None:
            os.environ.pop("TG_PACKED", None)
        else:
            os.environ["TG_PACKED"] = packed_before
    o = po.OracleBatch(B, W, H, rs, gravity, Q, action_ids=ids, rewards=rws, nthreads=8 if B > 1000 else 1)
    seeds = r.integers(1, 1 << 40, B).astype(np.uint64)
    e.reset(seeds); o.reset(seeds)
    same_state(e, o, numpy_rules, f"after reset [{tag}]")
    na = 4 * W if grouped else (8 if numpy_rules else 7)
    done, total = 0, int(steps or os.environ.get("FUZZ_STEPS", 260)) // (4 if B > 1000 else 1)
    while done < total:
        K = int(r.choice([1, 1, 3, 16, 40]))
        acts = r.integers(0, na, (K, B))
        if K == 1 and r.random() < 0.5:
            re_, te, le = e.step(acts[0].astype(np.int32))
            got = (None, re_[None], te[None], le[None])
            got_obs_last = e.observation() if obs != N.OBS_NONE else None
        else:
            if B % 4:  # (tg_rollout needs a batch that is a multiple of 4: single steps otherwise)
                res = [e.step(a.astype(np.int32)) for a in acts]
                got = (None, np.stack([x[0] for x in res]), np.stack([x[1] for x in res]), np.stack([x[2] for x in res]))
            else:
                got = e.rollout(acts.astype(np.int8), want_obs=obs in (N.OBS_BOARD, N.OBS_FN))
            got_obs_last = e.observation() if obs != N.OBS_NONE else None
        for k in range(K):
            if grouped:
                ro, to, lo, _ = o.grouped_step(acts[k].astype(np.int32), terminate_on_illegal=toi, autoreset=PO_AR[autoreset])
            else:
                ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=PO_AR[autoreset])
            where = f"at transition {done + k} [{tag}]"
            assert np.array_equal(got[2][k], to), f"terminated differs {where}"
            assert np.array_equal(got[3][k], lo), f"lines differ {where}"
            assert np.array_equal(got[1][k], ro), f"reward differs {where}: {got[1][k][:8]} vs {ro[:8]}"
            if got[0] is not None:
                same_obs(got[0][k], o.observe(PO_OBS[obs]), where)
        done += K
        if got_obs_last is not None:
            same_obs(got_obs_last, o.observe(PO_OBS[obs]), f"after {done} transitions [{tag}]")
        same_state(e, o, numpy_rules, f"after {done} transitions [{tag}]")
        if numpy_rules and r.random() < 0.3:  # the wrappers' views of the current state, from whatever handle this is
            w = f"after {done} transitions [{tag}]"
            same_obs(e.observe_as(N.OBS_DICT), o.observe(po.OBS_DICT), "observe_as(dict) " + w)
            same_obs(e.observe_as(N.OBS_RGB), o.observe(po.OBS_RGB), "observe_as(rgb) " + w)
            pl, cu = e.features_as(placement=True, current=True)
            assert np.array_equal(pl, o.grouped_features()) and np.array_equal(cu, o.features()), "features_as differ " + w
            if grouped:
                m, bds = e.placement_boards()
                mo, bo_ = o.grouped_enumerate(boards=True)
                assert np.array_equal(m, mo) and np.array_equal(bds, bo_), "placement boards / action mask differ " + w
        u = r.random()
        if u < 0.15:  # masked reset, some envs with a new seed (NUMPY: 0 keeps the running generator)
            mask = (r.random(B) < 0.3).astype(np.uint8)
            sd = np.where(r.random(B) < 0.5, r.integers(1, 1 << 40, B), 0).astype(np.uint64)
            if not numpy_rules:
                sd = r.integers(1, 1 << 40, B).astype(np.uint64)
            e.reset(sd, mask); o.reset(sd, mask)
            same_state(e, o, numpy_rules, f"after a masked reset at {done} [{tag}]")
        elif u < 0.33 and not (packed and not numpy_rules):
            # a caller-provided board: random cells, some rows completely full (play never leaves one: the next lock clears them,
            # envs/tetris.py:493), the active piece wherever it is — possibly on top of filled cells.  (Packed FUNCTIONAL handles
            # refuse the sums such an overlap makes: tests/test_gpu_parity.py covers that refusal.)
            bb = np.where(r.random((B, H, W)) < r.uniform(0.1, 0.6), r.integers(2, 9, (B, H, W)), 0).astype(np.uint8)
            bb[:, : H // 3] = 0
            for i in np.flatnonzero(r.random(B) < 0.5):
                for row in r.choice(np.arange(H // 3, H), size=int(r.integers(1, 4)), replace=False):
                    bb[i, row] = r.integers(2, 9, W)
            if H > 5:  # (round 6) a row of bedrock cells below the spawn rows: never cleared under the NUMPY rules (envs/tetris.py:493)
                for i in np.flatnonzero(r.random(B) < 0.25):
                    bb[i, int(r.integers(max(4, H // 3), H))] = 1
            e.set_state(board=bb); o.set_state(board=bb)
            same_state(e, o, numpy_rules, f"after a board upload at {done} [{tag}]")
        elif u < 0.43:  # the state through the host and back: nothing may change, the look-ahead is rebuilt
            st = e.get_state()
            e.set_state(*st)
            same_state(e, o, numpy_rules, f"after a get_state / set_state round trip at {done} [{tag}]")
        elif u < 0.53 and numpy_rules:
            # hand-made records (round 6; the reference's tests assign these attributes directly, tests/test_base_env/**): any piece /
            # rotation / position the reference's slicing tolerates — on filled cells, inside the bedrock —, holder, queue, bag, flags
            bb, rc, gg = o.get_state()
            sel = r.random(B) < 0.6
            nmat = np.array([4, 2, 3, 3, 3, 3, 3])
            pc = r.integers(0, 7, B)
            n = nmat[pc]
            rc["piece"][sel] = pc[sel]
            rc["rot"][sel] = r.integers(0, 4, B)[sel]
            rc["x"][sel] = (1 + r.integers(0, 1 << 30, B) % (W + 8 - n - 1))[sel]
            rc["y"][sel] = (r.integers(0, 1 << 30, B) % (H + 4 - n))[sel]
            rc["val"][sel] = pc[sel] + 2
            hp = np.where(r.random(B) < 0.5, r.integers(0, 7, B), 255)
            rc["holder_piece"][sel] = hp[sel]
            rc["holder_rot"][sel] = np.where(hp == 255, 0, r.integers(0, 4, B))[sel]
            rc["holder_val"][sel] = np.where(hp == 255, 0, hp + 2)[sel]
            rc["flags"][sel] = ((r.random(B) < 0.08) | ((r.random(B) < 0.4) << 1)).astype(np.uint8)[sel]
            rc["queue"][sel, :Q] = r.integers(0, 7, (B, Q))[sel]
            rc["bag"][sel] = np.stack([r.permutation(7) for _ in range(B)])[sel]
            rc["bag_index"][sel] = r.integers(0, 7, B)[sel]
            e.set_state(rec=rc); o.set_state(rec=rc)
            same_state(e, o, numpy_rules, f"after a record upload at {done} [{tag}]")
            if obs != N.OBS_NONE:
                same_obs(e.observation(), o.observe(PO_OBS[obs]), f"after a record upload at {done} [{tag}]")
        elif u < 0.53 and not packed:
            # ... and of the functional rule-set (engine == oracle only: the oracle's functional side is pinned by the reference's restated
            # known answers, not by vectors): any piece / rotation in its 4 x 4 box at x in [1, W + 3], y in [0, H - 1] — every position a
            # move may probe stays inside the padded board, where lax.dynamic_slice does not clamp (functional/core.py:86-100) —, bag, bag
            # index, game_over.  (Packed handles refuse the sums a piece on filled cells makes: tests/test_gpu_parity.py covers that.)
            bb, rc, gg = o.get_state()
            sel = r.random(B) < 0.6
            pc = r.integers(0, 7, B)
            rc["piece"][sel] = pc[sel]
            rc["rot"][sel] = r.integers(0, 4, B)[sel]
            rc["x"][sel] = r.integers(1, W + 4, B)[sel]
            rc["y"][sel] = r.integers(0, H, B)[sel]
            rc["val"][sel] = pc[sel] + 2
            rc["flags"][sel] = (r.random(B) < 0.08).astype(np.uint8)[sel]
            bag = np.zeros((B, 7), np.uint8)
            bag[:, :Q] = np.stack([r.permutation(Q) for _ in range(B)])
            rc["bag"][sel] = bag[sel]
            rc["bag_index"][sel] = r.integers(0, Q + 1, B)[sel]
            e.set_state(rec=rc); o.set_state(rec=rc)
            same_state(e, o, numpy_rules, f"after a record upload at {done} [{tag}]")
            if obs != N.OBS_NONE:
                same_obs(e.observation(), o.observe(PO_OBS[obs]), f"after a record upload at {done} [{tag}]")
    line = f"ok   {tag}: {done} transitions  [{e.step_kernel_name(fuse, grouped)}]"
    e.close()
    return line


if __name__ == "__main__":
    NCFG = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_start = time.time()
    sizes = draw_rtc_sizes(SEED)
    for ci in range(NCFG):
        print(run_config(SEED, ci, sizes), flush=True)
    print(f"{NCFG} configurations, seed {SEED}: {time.time() - t_start:.0f} s")
