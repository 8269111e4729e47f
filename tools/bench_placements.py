#!/usr/bin/env python3
"""Runs ON THE GPU BOX: throughput of the grouped placement-boards kernel (HBM-write bound: 4W*Hp*Wp bytes per env)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_gymnasium_amd import Engine, _native as N
for (W, H, Q, B) in ((10, 20, 4, 65536), (20, 40, 7, 16384)):
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, autoreset=N.AUTORESET_SAME_STEP)
    e.reset(np.arange(1, B + 1, dtype=np.uint64))
    for t in range(30):  # mid-game boards
        e.sample_legal_actions(seed=1, counter=t, fetch=False); e.step_sampled()
    nbytes = B * 4 * W * (H + 4) * (W + 8)
    d = e.dev_alloc(nbytes)
    for _ in range(3): e.placement_boards_device(d)
    e.sync(); e.timer_start()
    K = 20
    for _ in range(K): e.placement_boards_device(d)
    ms = e.timer_stop() / K
    print(f"placement boards {W}x{H}: {B} envs, {nbytes/1e6:.0f} MB per call, {ms*1e3:.1f} us = {nbytes/ms/1e6:.0f} GB/s written, "
          f"{B/ms*1e3/1e6:.1f} M envs/s")
    e.dev_free(d); e.close()
    # FeatureVectorObservation of all 4W placements + of the current board (tg_features: no boards are built) and the legal
    # mask alone (tg_grouped_enumerate(h, NULL)): one kernel each, device-resident results
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, features=True, autoreset=N.AUTORESET_SAME_STEP)
    e.reset(np.arange(1, B + 1, dtype=np.uint64))
    for t in range(30):
        e.sample_legal_actions(seed=1, counter=t, fetch=False); e.step_sampled()
    for name, call in (("features (4W placements x (W+3) + current)", lambda: e._chk(e.lib.tg_features(e.h), "tg_features")),
                       ("legal mask", lambda: e._chk(e.lib.tg_grouped_enumerate(e.h, None), "tg_grouped_enumerate"))):
        for _ in range(3): call()
        e.sync(); e.timer_start()
        for _ in range(K): call()
        ms = e.timer_stop() / K
        out_b = B * (4 * W + 1) * (W + 3) if name.startswith("features") else B * 4 * W
        in_b = B * W * H
        print(f"{name} {W}x{H}: {B} envs, {ms*1e3:.1f} us per call, {B/ms*1e3/1e6:.1f} M envs/s, "
              f"{(in_b + out_b)/ms/1e6:.0f} GB/s of algorithmic bytes ({in_b/1e6:.1f} MB boards read + {out_b/1e6:.1f} MB written)")
    e.close()
