#!/usr/bin/env python3
"""Runs ON THE GPU BOX: absolute timeline of ONE launch of the bits step kernel from the s_memtime stamps of a
-DTG_PHASE_TIMING build: per XCD (blockIdx % 8), when each role's waves start and end relative to the first stamp of
that XCD, and which role ends last (= the launch's critical path).  Extra -D flags are passed to the build."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, os.environ.get("TL_SRC", "tetris_gymnasium_amd/csrc"), "tg_engine.hip")  # TL_SRC: another source tree (_ab/<tag>)
out = "/tmp/libtg_phase.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DTG_PHASE_TIMING", "-DTG_STAMP_REALTIME",
                       "-DTG_FOR_EACH_SIZE(X)=X(10,20)", "-I" + os.path.join(ROOT, "tetris_gymnasium_amd", "csrc"), *sys.argv[1:], "-lhiprtc", "-shared", "-o", out, src])
os.environ.setdefault('TG_FUSE', '1')
os.environ.setdefault('TG_RNG_EVERY', '1')  # the stamped launch carries the RNG-role workgroups (they ride on every third launch otherwise)  # steady state of one transition per launch (a fused warm-up leaves an RNG backlog)
from tetris_gymnasium_amd import _native as N
N.LIB_PATH = out
from tetris_gymnasium_amd import Engine
B = 65536
fn = os.environ.get('TL_RULESET', 'numpy') == 'functional'  # TL_RULESET=functional: the functional rule-set's launch
e = Engine(B, ruleset=(N.RULESET_FUNCTIONAL if fn else N.RULESET_NUMPY), obs_mode=(N.OBS_FN if fn else {'board': N.OBS_BOARD, 'dict': N.OBS_DICT, 'rgb': N.OBS_RGB}[os.environ.get('TL_OBS', 'board')]), autoreset=N.AUTORESET_SAME_STEP)  # TL_OBS=dict / rgb
e.reset(np.arange(1, B + 1, dtype=np.uint64))
acts = np.random.default_rng(0).integers(0, 7 if fn else 8, (80, B)).astype(np.int8)
d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
e.step_k_device(d, 72); e.sync()
# launch time of the stamped build (HIP events over 7 launches) and then the one launch whose stamps are read
e.timer_start(); e.step_k_device(d + 72 * B, 7); ms = e.timer_stop()
e.step_k_device(d + 79 * B, 1); e.sync()
print(f"stamped build: {ms / 7 * 1e3:.2f} us per launch")
nb = B // 64
nr = B // int(os.environ.get('TL_RNG_ENVS', '128'))  # RNG-role workgroups (kRngEnvs envs each)
buf = np.zeros((B // 16, 16), np.uint64)
e.lib.tg_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert e.lib.tg_debug_phase_times(e.h, buf.ctypes.data_as(C.c_void_p), B // 16) == 0
tl = buf[:nb, 0:8].astype(np.int64)    # logic wave stamps 0..7
ts = buf[:nb, 8:13].astype(np.int64)   # stream wave stamps 0..4
rb = buf.reshape(-1)[nb * 16: nb * 16 + nr * 4].reshape(nr, 2, 2).astype(np.int64)  # [rng block][wave][start, end]
if fn or not rb.any():  # (FUNCTIONAL launches have no RNG-role workgroups: their slots stay zero — give them the step waves' range)
    rb = np.stack([np.stack([tl[:nr, 0], tl[:nr, 0]], -1), np.stack([tl[:nr, 0], tl[:nr, 0]], -1)], 1).astype(np.int64) if nr <= nb else \
        np.broadcast_to(np.array([tl[0, 0], tl[0, 0]], np.int64), (nr, 2, 2)).copy()
if os.environ.get('TL_DEBUG'):
    print('min/max per logic stamp', tl.min(axis=0), tl.max(axis=0)); print('stream', ts.min(axis=0), ts.max(axis=0)); print('rng', rb.min(), rb.max(), (rb == 0).sum())
step_blk = np.arange(nb)        # blockIdx of step workgroup i (the step workgroups lead the grid)
rng_blk = np.arange(nr) + nb
names_l = ["start", "loads done", "window read", "decided", "lock done", "published", "stores done", "end"]
names_s = ["start", "slab in LDS", "barrier passed", "write-back done", "end"]
for x in range(8):
    ls, ss, rr = tl[step_blk % 8 == x], ts[step_blk % 8 == x], rb[rng_blk % 8 == x]
    t0 = min(ls[:, 0].min(), ss[:, 0].min(), rr[:, :, 0].min())
    ends = {"rng": rr[:, :, 1].max() - t0, "logic": ls[:, 7].max() - t0, "stream": ss[:, 4].max() - t0}
    if x == 0:
        print("XCD 0 detail (cycles after the XCD's first stamp):  mean / max")
        for i, nm in enumerate(names_l):
            print(f"  logic  {nm:16s} {(ls[:, i] - t0).mean():8.0f} {(ls[:, i] - t0).max():8.0f}")
        for i, nm in enumerate(names_s):
            print(f"  stream {nm:16s} {(ss[:, i] - t0).mean():8.0f} {(ss[:, i] - t0).max():8.0f}")
        print(f"  rng    start            {(rr[:, :, 0] - t0).mean():8.0f} {(rr[:, :, 0] - t0).max():8.0f}")
        print(f"  rng    end              {(rr[:, :, 1] - t0).mean():8.0f} {(rr[:, :, 1] - t0).max():8.0f}")
    last = max(ends, key=ends.get)
    print(f"XCD {x}: span {max(ends.values()):6d} cycles; last wave to end: {last};  ends rng {ends['rng']} logic {ends['logic']} stream {ends['stream']}"
          f";  step WG start mean {(ss[:, 0] - t0).mean():.0f} max {(ss[:, 0] - t0).max()}")
# which workgroups end last?  (stream wave end vs launch position)
t00 = min(tl[:, 0].min(), ts[:, 0].min(), rb[:, :, 0].min())
t_end = max(tl[:, 7].max(), ts[:, 4].max(), rb[:, :, 1].max())
print(f"launch span on the device (first stamp of any wave -> last stamp of any wave, 100 MHz counter): {(t_end - t00) / 100:.2f} us "
      f"(the stamped build itself runs {ms / 7 * 1e3:.2f} us per launch by HIP events: span + the launch boundary)")
end = ts[:, 4] - t00
start = ts[:, 0] - t00
order = np.argsort(end)
print("stream-wave end by workgroup index decile (mean start / mean end / max end, 10 ns ticks):")
for d in range(10):
    sl = slice(d * nb // 10, (d + 1) * nb // 10)
    print(f"  WGs {sl.start:4d}-{sl.stop - 1:4d}: start {start[sl].mean():6.1f}  end {end[sl].mean():6.1f}  max {end[sl].max():5d}")
print("the 16 last workgroups to end:", sorted(order[-16:].tolist()), " their XCDs:", sorted((order[-16:] % 8).tolist()))
