#!/usr/bin/env python3
"""Runs ON THE GPU BOX: build an engine variant with extra -D flags into /tmp and bench it (measurement-only variants)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = sys.argv[1:]
out = "/tmp/libtg_exp.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DTG_FOR_EACH_SIZE(X)=X(10,20)", *(["-DTG_LEGACY_KERNELS"] if os.environ.get("TG_KERNEL", "bits") != "bits" else []), *flags, "-lhiprtc", "-shared", "-o", out,
                       os.path.join(ROOT, "tetris_gymnasium_amd", "csrc", "tg_engine.hip")])
from tetris_gymnasium_amd import _native as N
N.LIB_PATH = out
from tetris_gymnasium_amd import Engine
B, K = 65536, 512
e = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
e.reset(np.arange(1, B + 1, dtype=np.uint64))
acts = np.random.default_rng(0).integers(0, 8, (K, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
e.step_k_device(d, 64); e.sync()
best = []
for rep in range(5):
    e.timer_start(); e.step_k_device(d, K); ms = e.timer_stop()
    best.append(ms / K * 1e3)
print(f"flags={flags} kernel={os.environ.get('TG_KERNEL','bits')}: " + " ".join(f"{b:.2f}" for b in best) + " us/step")
