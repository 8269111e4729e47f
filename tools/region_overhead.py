"""What a short timed region costs beyond its kernels: wall clock of [sync; K one-transition launches; sync] against K
(median of `trials`), the time the enqueue call takes, and the same region by HIP events.  Intercept = fixed cost of a
region (first dispatch after idle + completion detection), slope = per launch.  usage: python tools/region_overhead.py"""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402

B = int(os.environ.get("AB_BATCH", 65536))
trials = int(os.environ.get("AB_TRIALS", 60))
e = Engine(B, 10, 20, ruleset=N.RULESET_NUMPY, queue_size=4, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP,
           fuse=int(os.environ.get("AB_FUSE", 1)))  # (1: one transition per launch, the headline configuration)
e.reset(np.arange(B, dtype=np.uint64))
T = 128
acts = np.random.default_rng(0).integers(0, 8, (T, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes)
e.dev_upload(d, acts)
e.step_k_device(d, 64)
e.sync()
rows = []
for K in (1, 2, 5, 10, 20, 40, 80):
    wall, enq, ev = [], [], []
    for t in range(trials):
        e.sync()
        t0 = time.perf_counter()
        e.step_k_device(d, K)
        t1 = time.perf_counter()
        e.sync()
        t2 = time.perf_counter()
        wall.append((t2 - t0) * 1e6)
        enq.append((t1 - t0) * 1e6)
    for t in range(trials // 2):
        e.sync()
        e.timer_start()
        e.step_k_device(d, K)
        ev.append(e.timer_stop() * 1e3)
    rows.append((K, statistics.median(wall), min(wall), statistics.median(enq), statistics.median(ev)))
    print(f"K={K:3d}  wall median {rows[-1][1]:7.1f} us  min {rows[-1][2]:7.1f}  enqueue call {rows[-1][3]:6.1f} us  HIP events {rows[-1][4]:7.1f} us"
          f"  -> per launch {rows[-1][1] / K:6.2f} (wall) {rows[-1][4] / K:6.2f} (events)", flush=True)
ks = np.array([r[0] for r in rows], float)
for name, col in (("wall", 1), ("events", 4)):
    ys = np.array([r[col] for r in rows])
    slope, icpt = np.polyfit(ks, ys, 1)
    print(f"{name}: {slope:.2f} us per launch + {icpt:.1f} us per region")
