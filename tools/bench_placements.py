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
