#!/usr/bin/env python3
"""Runs ON THE GPU BOX: long parity soak against the C oracle (test infrastructure): tens of thousands of transitions per
env so that the mod-128 ring counters wrap many times, across launch kinds (single steps, 1 / 3 / 8 transitions per launch,
trajectories) and configurations.  usage: python tests/soak_parity.py [steps]   (not collected by pytest: minutes of oracle time)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from tetris_gymnasium_amd import _native as N

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
MAPPED = {1, 6}  # cases that run with a non-stock ActionsMapping / RewardsMapping (a permutation of the ids; round 5: per-handle data)
CASES = [  # (W, H, Q, B, grouped, fuse, ruleset)
    (10, 20, 4, 320, False, 1, N.RULESET_NUMPY), (10, 20, 4, 200, False, 16, N.RULESET_NUMPY), (10, 20, 7, 132, True, 3, N.RULESET_NUMPY),
    (20, 40, 7, 100, True, 1, N.RULESET_NUMPY), (6, 12, 2, 256, False, 8, N.RULESET_NUMPY), (10, 20, 7, 192, False, 16, N.RULESET_FUNCTIONAL),
    (7, 14, 4, 168, False, 1, N.RULESET_NUMPY), (7, 14, 7, 100, False, 8, N.RULESET_FUNCTIONAL),  # 98-byte slabs, ragged batches
]
for ci, (W, H, Q, B, grouped, fuse, rs) in enumerate(CASES):
    os.environ["TG_FUSE"] = str(fuse)
    from tetris_gymnasium_amd import Engine
    numpy_rules = rs == N.RULESET_NUMPY
    obs = N.OBS_BOARD if numpy_rules else N.OBS_FN
    surf = os.environ.get("SOAK_OBS", "")  # SOAK_OBS=dict / rgb: the NUMPY cases with that observation packed in the step kernel
    if numpy_rules and not grouped and surf in ("dict", "rgb"):
        obs = N.OBS_DICT if surf == "dict" else N.OBS_RGB
    ids, rws = ((3, 7, 0, 5, 1, 2, 4, 6), (0.5, 1.0, -5.25, -0.25)) if ci in MAPPED else (None, None)
    e = Engine(B, W, H, ruleset=rs, queue_size=Q, obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP, grouped=grouped,
               actions_mapping=ids, rewards_mapping=rws)
    o = po.OracleBatch(B, W, H, rs, True, Q, action_ids=ids, rewards=rws)
    seeds = np.arange(11, 11 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(W * 100 + fuse)
    na = 4 * W if grouped else (8 if numpy_rules else 7)
    done, CH = 0, 500
    while done < STEPS:
        K = min(CH, STEPS - done)
        acts = rng.integers(0, na, (K, B)).astype(np.int8)
        d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts); e.step_k_device(d, K); e.dev_free(d)
        for k in range(K):
            if grouped: o.grouped_step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            else: o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        done += K
        be, re_, ge = e.get_state(); bo, ro, go = o.get_state()
        assert np.array_equal(be, bo), f"board differs after {done} steps"
        for f in re_.dtype.names:
            if f != "reserved": assert np.array_equal(re_[f], ro[f]), f"rec.{f} differs after {done} steps"
        if numpy_rules:
            for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
                assert np.array_equal(ge[f], go[f]), f"rng.{f} differs after {done} steps"
        po_obs = {N.OBS_BOARD: po.OBS_BOARD, N.OBS_FN: po.OBS_FN, N.OBS_DICT: po.OBS_DICT, N.OBS_RGB: po.OBS_RGB}[obs]
        got, ref = e.observation(), o.observe(po_obs)
        if isinstance(got, dict):
            for kk in got:
                assert np.array_equal(got[kk], ref[kk]), f"observation[{kk}] differs after {done} steps"
        else:
            assert np.array_equal(got, ref), f"observation differs after {done} steps"
    print(f"ok  {W}x{H} Q={Q} B={B} grouped={grouped} fuse={fuse} ruleset={rs} obs={obs}{' custom mappings' if ids else ''}: {STEPS} transitions per env  [{e.step_kernel_name(fuse, grouped)}]")
    e.close()
