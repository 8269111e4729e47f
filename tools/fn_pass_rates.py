#!/usr/bin/env python3
"""CPU (oracle) measurement behind DESIGN §3b "why the functional look-ahead pass cannot be skipped": per 64-env wave and
transition under the uniform random policy (10x20, SAME_STEP autoreset), how many lanes lock, lock with a stack within reach of
the spawn rows (the reset bag S2 must be sorted ahead), lock with the 7-bag used up (the refill bag Q0 must be sorted), reset —
and in which fraction of the wave-transitions NO lane has any such need (the only transitions a wave-uniform skip could use).
Also: how often the threefry LEVEL could be skipped if keys were extended only when a lane's chain runs dry (depth 0), for
chains up to 3 / 5 / 7 keys deep — ignoring that a tall-stack lock needs its chain at full depth for S2.
Usage: python tools/fn_pass_rates.py   (about a minute)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

B, T, WARM = 64 * 64, 400, 100
o = po.OracleBatch(B, 10, 20, po.RULESET_FUNCTIONAL, nthreads=os.cpu_count() or 1)
o.reset(np.arange(1, B + 1, dtype=np.uint64))
rng = np.random.default_rng(0)
keys = ("lock", "tall_lock", "refill_lock", "reset")
tot = dict.fromkeys(keys, 0)
anyw = dict.fromkeys(keys, 0)
idle = 0
prev = o.get_state()[1]["bag_index"].copy()
n = 0
lock_hist = []
for t in range(T):
    rew, term, lines = o.step(rng.integers(0, 7, B), autoreset=po.AUTORESET_SAME_STEP)
    board, rec, _ = o.get_state()
    term = term.astype(bool)
    bidx = rec["bag_index"]
    ev = {"lock": (bidx != prev) | term, "reset": term}  # (a same-step reset happens behind a lock)
    ev["tall_lock"] = ev["lock"] & ~term & board[:, 0:6, :].any(axis=(1, 2))
    ev["refill_lock"] = ev["lock"] & ~term & (bidx >= 7)
    prev = bidx.copy()
    if t < WARM:
        continue
    n += 1
    lock_hist.append(ev["lock"].reshape(-1, 64))
    for k in keys:
        tot[k] += int(ev[k].sum())
        anyw[k] += int(ev[k].reshape(-1, 64).any(1).sum())
    idle += int((~(ev["tall_lock"] | ev["refill_lock"] | ev["reset"]).reshape(-1, 64).any(1)).sum())
W = n * B // 64
print(f"{W} wave-transitions (64 envs per wave, uniform random actions, 10x20, same-step autoreset)")
for k in keys:
    print(f"  {k:12s} {tot[k] / W:6.2f} lanes per wave-transition; some lane in {100.0 * anyw[k] / W:5.1f} % of them")
print(f"  no lane needs a sorted bag or a fresh chain (tall lock / refill lock / reset): {100.0 * idle / W:.1f} % of the wave-transitions")
locks = np.stack(lock_hist)  # [n][waves][64]
for cap in (3, 5, 7):
    d = np.full(locks.shape[1:], cap)
    passes = 0
    for t in range(locks.shape[0]):
        d = d - locks[t]
        need = (d <= 0).any(1)
        passes += int(need.sum())
        d = np.where(need[:, None], np.minimum(d + 1, cap), d)
        d = np.maximum(d, 0)
    print(f"  key chains up to {cap} deep, extended only when some lane's chain runs dry: the level runs in "
          f"{100.0 * passes / (locks.shape[0] * locks.shape[1]):.0f} % of the wave-transitions")
