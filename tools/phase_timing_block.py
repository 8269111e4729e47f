#!/usr/bin/env python3
"""Runs ON THE GPU BOX: phase stamps (s_memtime) of the block kernel: P0 load, P1 fast path, P2 slow path, P3 stores."""
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
acts = np.random.default_rng(0).integers(0, 8, (80, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
e.step_k_device(d, 79); e.sync(); e.step_k_device(d + 79 * B, 1); e.sync()
nb = B // (16 * int(os.environ.get("TG_BLOCK_WAVES", "4")))
buf = np.zeros((B // 16, 16), np.uint64)
e.lib.tg_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert e.lib.tg_debug_phase_times(e.h, buf.ctypes.data_as(C.c_void_p), B // 16) == 0
t = buf[:nb, 10:15].astype(np.int64)
for i, nm in enumerate(["P0 load+stage", "P1 fast path + worklist", "P2 slow path (dense)", "P3 write-back, stamp, obs, records"]):
    print(f"  {nm:36s} mean {np.diff(t, axis=1)[:, i].mean():8.1f} ticks   max {np.diff(t, axis=1)[:, i].max():8.1f}")
print("block lifetime mean", (t[:, 4] - t[:, 0]).mean(), "max", (t[:, 4] - t[:, 0]).max(), "; kernel span", t[:, 4].max() - t[:, 0].min())
# inside P2 (stamps of lanes_logic, written by lane 0 of whichever wave ran last): relative to P2 start
tt = buf[:nb].astype(np.int64)
names = {2: "swap/LUT done", 3: "collide+decide done", 4: "drop+stamp done", 5: "rows+clear done", 6: "spawn(+shuffle) done", 7: "terminate flags", 8: "reset: zero slab+rare resolve", 9: "reset: shuffle", 13: "reset draws + end (P2 end)"}
prev = tt[:, 12]
for k in (2, 3, 4, 5, 6, 7, 8, 9, 13):
    cur = np.where(tt[:, k] >= tt[:, 12], tt[:, k], prev)
    print(f"  P2 -> {names[k]:24s} +{(cur - prev).mean():8.1f} ticks")
    prev = cur
