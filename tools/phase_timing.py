#!/usr/bin/env python3
"""Runs ON THE GPU BOX: builds a TG_PHASE_TIMING variant of the engine into /tmp, steps it and prints the mean
s_memtime delta (100 MHz constant clock ticks → ns) between the phase stamps of the lane kernel."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "tetris_gymnasium_amd", "csrc", "tg_engine.hip")
out = "/tmp/libtg_phase.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DTG_PHASE_TIMING", "-DTG_LEGACY_KERNELS", "-lhiprtc", "-shared", "-o", out, src])
from tetris_gymnasium_amd import _native as N
N.LIB_PATH = out
from tetris_gymnasium_amd import Engine
B = 65536
e = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
e.reset(np.arange(1, B + 1, dtype=np.uint64))
rng = np.random.default_rng(0)
acts = rng.integers(0, 8, (80, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
e.step_k_device(d, 79); e.sync()
e.step_k_device(d + 79 * B, 1); e.sync()
EPW = 16 if os.environ.get("TG_KERNEL", "quad") == "quad" else 64
nw = B // EPW
buf = np.zeros((nw, 16), np.uint64)
e.lib.tg_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc = e.lib.tg_debug_phase_times(e.h, buf.ctypes.data_as(C.c_void_p), nw)
assert rc == 0, rc
t = buf[:, :10].astype(np.int64)
for i in range(1, 10):  # stamps inside skipped blocks stay 0: forward-fill
    t[:, i] = np.where(t[:, i] == 0, t[:, i - 1], t[:, i])
names = ["A load+stage", "unpack+LUT+swap", "cells+collide+decide", "lock: drop+stamp", "rows+clear loop", "spawn(+shuffle)", "E reset", "F writeback+obs", "rec/out stores", "-"]
dt = np.diff(t, axis=1)
print("per-wave mean shader cycles (s_memtime) per phase, kernel", os.environ.get("TG_KERNEL", "quad"))
for i in range(9):
    col = dt[:, i]
    ok = (t[:, i] > 0) & (t[:, i + 1] > 0)
    print(f"  {i}->{i+1} {names[i]:22s} mean {col[ok].mean():8.1f} cycles   (n={ok.sum()})")
print("wave lifetime mean", (t[:, 9] - t[:, 0]).mean(), "cycles; kernel span", (t[:, 9].max() - t[:, 0].min()), "cycles")
