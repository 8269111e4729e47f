#!/usr/bin/env python3
"""GPU experiment: step time at batch 65536 for fixed action patterns (which paths cost what)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_gymnasium_amd import Engine, _native as N
B, K = 65536, 256
for name, gen in (("uniform 0..7", lambda r: r.integers(0, 8, (K, B))), ("all no-op (gravity locks only)", lambda r: np.full((K, B), 7)),
                  ("all hard drop", lambda r: np.full((K, B), 5)), ("left/right only", lambda r: r.integers(0, 2, (K, B))),
                  ("no hard drop, no swap", lambda r: r.choice([0, 1, 2, 3, 4, 7], (K, B)))):
    e = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
    e.reset(np.arange(1, B + 1, dtype=np.uint64))
    acts = gen(np.random.default_rng(0)).astype(np.int8)
    d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
    e.step_k_device(d, 64); e.sync()
    e.timer_start(); e.step_k_device(d, K); ms = e.timer_stop()
    print(f"{name:32s}: {ms / K * 1e3:7.2f} us/step")
    e.dev_free(d); e.close()
