#!/usr/bin/env python3
"""GPU experiment: one engine with B envs vs. S engines with B/S envs each on separate streams (same GPU)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_gymnasium_amd import Engine, _native as N
B, K = 65536, 512
rng = np.random.default_rng(0)
for S in (1, 2, 4, 8):
    engs, ptrs = [], []
    for k in range(S):
        b = B // S
        e = Engine(b, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, env_offset=k * b)
        e.reset(np.arange(1 + k * b, 1 + (k + 1) * b, dtype=np.uint64))
        acts = rng.integers(0, 8, (K, b)).astype(np.int8)
        d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
        engs.append(e); ptrs.append(d)
    for e, d in zip(engs, ptrs): e.step_k_device(d, 64)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    # interleave launches so the streams progress together
    CH = 32
    for c in range(0, K, CH):
        for e, d in zip(engs, ptrs): e.step_k_device(d + c * e.B, CH)
    for e in engs: e.sync()
    dt = time.perf_counter() - t0
    print(f"streams={S}: {B*K/dt:.3e} env steps/s, {dt/K*1e6:.2f} us per full-batch step")
    for e in engs: e.close()
