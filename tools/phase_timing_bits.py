#!/usr/bin/env python3
"""Runs ON THE GPU BOX: s_memtime phase stamps of the bits kernel (one wave = 64 envs per workgroup)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "tetris_gymnasium_amd", "csrc", "tg_engine.hip")
out = "/tmp/libtg_phase.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DTG_PHASE_TIMING",
                       "-DTG_FOR_EACH_SIZE(X)=X(10,20)", *sys.argv[1:], "-lhiprtc", "-shared", "-o", out, src])
os.environ.setdefault('TG_FUSE', '1')
os.environ.setdefault('TG_RNG_EVERY', '1')  # the stamped launch carries the RNG-role workgroups (they ride on every third launch otherwise)  # steady state of one transition per launch (a fused warm-up leaves an RNG backlog)
from tetris_gymnasium_amd import _native as N
N.LIB_PATH = out
from tetris_gymnasium_amd import Engine
B = 65536
FN = os.environ.get('PT_RULESET') == 'functional'
e = Engine(B, ruleset=(N.RULESET_FUNCTIONAL if FN else N.RULESET_NUMPY), obs_mode=N.OBS_FN if FN else {'board': N.OBS_BOARD, 'dict': N.OBS_DICT, 'rgb': N.OBS_RGB}[os.environ.get('PT_OBS', 'board')], autoreset=N.AUTORESET_SAME_STEP)
e.reset(np.arange(1, B + 1, dtype=np.uint64))
acts = np.random.default_rng(0).integers(0, 7 if os.environ.get('PT_RULESET') == 'functional' else 8, (80, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
KL = int(os.environ.get('TG_LAST_K', '1'))
e.step_k_device(d, 80 - KL); e.sync(); e.step_k_device(d + (80 - KL) * B, KL); e.sync()
nb = B // 64
buf = np.zeros((B // 16, 16), np.uint64)
e.lib.tg_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert e.lib.tg_debug_phase_times(e.h, buf.ctypes.data_as(C.c_void_p), B // 16) == 0
tl = buf[:nb, 0:8].astype(np.int64)    # logic wave
ts = buf[:nb, 8:13].astype(np.int64)   # stream wave
def show(t, names):
    dd = np.diff(t, axis=1)
    for i, nm in enumerate(names):
        print(f"  {nm:44s} mean {dd[:, i].mean():8.1f}   p50 {np.median(dd[:, i]):8.1f}   max {dd[:, i].max():8.1f}")
rb = buf.reshape(-1)[nb * 16: nb * 16 + (B // 128) * 4].reshape(-1, 2).astype(np.int64)
rl = rb[:, 1] - rb[:, 0]
print(f"RNG waves ({len(rl)}): lifetime mean {rl.mean():.0f} p50 {np.median(rl):.0f} max {rl.max()}")
print("logic wave:")
show(tl, ["A small loads + cols->LDS", "B0 unpack/swap + window read", "B1 tests/decide", "B2 lock path",
          "B3 reset path + publish", "barrier + D rec/out/cols stores", "E refresh (PCG)"])
print("stream wave:")
show(ts, ["slab loads -> LDS", "wait for logic (barrier)", "lock stamp / clear / reset / write-back", "obs stamp + obs stores"])
ll, ls = tl[:, 7] - tl[:, 0], ts[:, 4] - ts[:, 0]
print("logic wave lifetime: mean", ll.mean(), "p50", np.median(ll), "max", ll.max())
print("stream wave lifetime: mean", ls.mean(), "p50", np.median(ls), "max", ls.max())
