#!/usr/bin/env python3
"""Runs ON THE GPU BOX: where the two waves of every step workgroup ran (HW_ID / XCC_ID read by the stamped build):
how many logic waves share a SIMD, how the workgroups spread over the CUs."""
import ctypes as C, collections, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "tetris_gymnasium_amd", "csrc", "tg_engine.hip")
out = "/tmp/libtg_place.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DTG_PHASE_TIMING",
                       "-DTG_FOR_EACH_SIZE(X)=X(10,20)", *sys.argv[1:], "-lhiprtc", "-shared", "-o", out, src])
os.environ.setdefault("TG_FUSE", "1")
os.environ.setdefault('TG_RNG_EVERY', '1')  # the stamped launch carries the RNG-role workgroups (they ride on every third launch otherwise)
from tetris_gymnasium_amd import _native as N
N.LIB_PATH = out
from tetris_gymnasium_amd import Engine
B = int(os.environ.get("WP_BATCH", "65536"))
e = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
e.reset(np.arange(1, B + 1, dtype=np.uint64))
acts = np.random.default_rng(0).integers(0, 8, (40, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
e.step_k_device(d, 40); e.sync()
nb = B // 64
buf = np.zeros((B // 16, 16), np.uint64)
e.lib.tg_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert e.lib.tg_debug_phase_times(e.h, buf.ctypes.data_as(C.c_void_p), B // 16) == 0
hl, hs, xcc = buf[:nb, 13].astype(np.int64), buf[:nb, 14].astype(np.int64), buf[:nb, 15].astype(np.int64) & 0xF
def fields(h):
    return {"wave": h & 0xF, "simd": (h >> 4) & 3, "pipe": (h >> 6) & 3, "cu": (h >> 8) & 0xF, "sh": (h >> 12) & 1, "se": (h >> 13) & 7}
fl, fs = fields(hl), fields(hs)
cu_key = lambda f, i: (int(xcc[i]), int(f["se"][i]), int(f["sh"][i]), int(f["cu"][i]))
per_cu = collections.defaultdict(list)
for i in range(nb):
    per_cu[cu_key(fl, i)].append(("L", int(fl["simd"][i]), i))
    per_cu[cu_key(fs, i)].append(("S", int(fs["simd"][i]), i))
print(f"{nb} workgroups on {len(per_cu)} CUs; workgroups per CU: {collections.Counter(len(v) // 2 for v in per_cu.values())}")
same_cu = sum(cu_key(fl, i) == cu_key(fs, i) for i in range(nb))
print(f"logic and stream wave on the same CU: {same_cu} of {nb};  on the same SIMD: {int(((fl['simd'] == fs['simd'])).sum())}")
lpersimd = collections.Counter()
for k, v in per_cu.items():
    c = collections.Counter(s for r, s, _ in v if r == "L")
    for s in range(4):
        lpersimd[c.get(s, 0)] += 1
print("logic waves per SIMD (count of SIMDs):", dict(sorted(lpersimd.items())))
for k in list(sorted(per_cu))[:6]:
    print(k, sorted((r, s, i) for r, s, i in per_cu[k]))
print("first workgroups -> (xcc, se, sh, cu, simdL, simdS):", [(cu_key(fl, i) + (int(fl['simd'][i]), int(fs['simd'][i]))) for i in range(12)])
