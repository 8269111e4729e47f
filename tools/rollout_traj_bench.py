#!/usr/bin/env python3
"""Runs ON THE GPU BOX: tg_step_k (every transition's observation lands in the SAME buffer) against tg_rollout (each
transition has its own slot of a trajectory buffer), 8 transitions per launch, device-resident actions."""
import ctypes as C, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TG_FUSE", "8")
from tetris_gymnasium_amd import _native as N
from tetris_gymnasium_amd import Engine
B, K = 65536, 256
e = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
e.reset(np.arange(1, B + 1, dtype=np.uint64))
acts = np.random.default_rng(0).integers(0, 8, (K, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
d_o = e.dev_alloc(K * B * 200); d_r = e.dev_alloc(K * B * 8)
e.step_k_device(d, 64); e.sync()
for name, fn in (("tg_step_k ", lambda: e.step_k_device(d, K)),
                 ("tg_rollout", lambda: e._chk(e.lib.tg_rollout(e.h, C.c_void_p(d), K, C.c_void_p(d_o), C.c_void_p(d_r)), "tg_rollout"))):
    ts = []
    for _ in range(5):
        e.timer_start(); fn(); ts.append(e.timer_stop() / K * 1e3)
    print(f"{name}: median {statistics.median(ts):.2f} us per transition   " + " ".join(f"{t:.2f}" for t in ts))
