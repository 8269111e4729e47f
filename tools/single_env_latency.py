"""Where the time of one single-env step goes (config 1: one env per handle, dict observation): the pieces of the one call
`Tetris.step` makes (tg_step_fetch_many: action upload, launch, copies, wait), each timed alone.  usage: python tools/single_env_latency.py"""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402
from tetris_gymnasium_amd.envs.tetris import Tetris  # noqa: E402

n = int(os.environ.get("AB_STEPS", 3000))


def med(f, warm=200):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts), min(ts)


env = Tetris()
env.reset(seed=1)
rng = np.random.default_rng(0)
acts = rng.integers(0, 7, 100000)
it = iter(range(10**9))


def full():
    _, _, term, _, _ = env.step(int(acts[next(it) % 100000]))
    if term:
        env.reset()


print("Tetris.step (whole Python call)            median %.1f  min %.1f us" % med(full))

e = Engine(1, 10, 20, ruleset=N.RULESET_NUMPY, queue_size=4, obs_mode=N.OBS_DICT, autoreset=N.AUTORESET_SAME_STEP)
e.reset(np.array([1], np.uint64))
a1 = np.zeros(1, np.int32)
lib = e.lib
hs = (C.c_void_p * 1)(e.h.value)
ap = (C.c_void_p * 1)(a1.ctypes.data)
obs_b = e.device_ptr(N.BUF_OBS)[1]
out_b = e.device_ptr(N.BUF_STEP_OUT)[1]
pin = e.pinned_bytes(obs_b + out_b + 64)
which = (C.c_int * 2)(N.BUF_OBS, N.BUF_STEP_OUT)
dst = (C.c_void_p * 2)(pin.ctypes.data, pin.ctypes.data + (obs_b + 63) // 64 * 64)
nb = (C.c_size_t * 2)(obs_b, out_b)
d_act = e.dev_alloc(64)
e.dev_upload(d_act, np.zeros(16, np.int32))
dp = (C.c_void_p * 1)(d_act)

print("tg_sync on an idle stream                  median %.1f  min %.1f us" % med(lambda: e.sync()))
print("tg_step (host action) + tg_sync            median %.1f  min %.1f us" % med(lambda: (e.step_async(a1), e.sync())))
print("tg_step (device action) + tg_sync          median %.1f  min %.1f us" % med(lambda: (e.step_device(d_act), e.sync())))
print("tg_step_fetch_many device action, 0 copies median %.1f  min %.1f us" % med(lambda: (lib.tg_step_fetch_many(1, hs, dp, N.MEM_DEVICE, 0, None, None, None, None), e.sync())))
nb1 = (C.c_size_t * 2)(obs_b, 0)
print("tg_step_fetch_many device action, 1 copy   median %.1f  min %.1f us" % med(lambda: lib.tg_step_fetch_many(1, hs, dp, N.MEM_DEVICE, 2, which, dst, nb1, None)))
print("tg_step_fetch_many device action, 2 copies median %.1f  min %.1f us" % med(lambda: lib.tg_step_fetch_many(1, hs, dp, N.MEM_DEVICE, 2, which, dst, nb, None)))
print("tg_step_fetch_many host action, 2 copies   median %.1f  min %.1f us" % med(lambda: lib.tg_step_fetch_many(1, hs, ap, N.MEM_HOST, 2, which, dst, nb, None)))
pa = e.pinned_bytes(4).view(np.int32)
pa[0] = 0
pap = (C.c_void_p * 1)(pa.ctypes.data)
print("... page-locked host action, 2 copies      median %.1f  min %.1f us" % med(lambda: lib.tg_step_fetch_many(1, hs, pap, N.MEM_HOST, 2, which, dst, nb, None)))
gap = e.device_ptr(N.BUF_STEP_OUT)[0] - e.device_ptr(N.BUF_OBS)[0]
pin2 = e.pinned_bytes(gap + out_b)
dst2 = (C.c_void_p * 2)(pin2.ctypes.data, pin2.ctypes.data + gap)
print("... device action, copies joined (gap %d)  median %.1f  min %.1f us" % ((gap,) + med(lambda: lib.tg_step_fetch_many(1, hs, dp, N.MEM_DEVICE, 2, which, dst2, nb, None))))
print("... page-locked action, copies joined      median %.1f  min %.1f us" % med(lambda: lib.tg_step_fetch_many(1, hs, pap, N.MEM_HOST, 2, which, dst2, nb, None)))
print("obs bytes", obs_b, "out bytes", out_b)

# the same question for the vector adapter at the sizes RL scripts use (gymnasium contract: NumPy results, copy=True)
from tetris_gymnasium_amd.vector import TetrisVectorEnv  # noqa: E402

for nenv in (16, 256, 4096):
    for kind in ("board", "dict"):
        v = TetrisVectorEnv(nenv, observation=kind, autoreset_mode="same_step")
        v.reset(seed=1)
        a = rng.integers(0, 8, (64, nenv)).astype(np.int32)
        k = iter(range(10**9))
        hold = [None]

        def vstep():
            hold[0] = v.step(a[next(k) % 64])

        print("TetrisVectorEnv(%5d, %-5s).step             median %.1f  min %.1f us" % ((nenv, kind) + med(vstep)))
        v.close()

# the single-env wrappers the reference's training examples use (examples/train_lin_grouped.py: grouped actions + feature vectors)
from tetris_gymnasium_amd.wrappers.grouped import GroupedActionsObservations  # noqa: E402
from tetris_gymnasium_amd.wrappers.observation import FeatureVectorObservation, RgbObservation  # noqa: E402

n = max(200, n // 10)
for label, mk in (("GroupedActionsObservations(Tetris())", lambda: GroupedActionsObservations(Tetris())),
                  ("... with [FeatureVectorObservation]", lambda: (lambda t: GroupedActionsObservations(t, observation_wrappers=[FeatureVectorObservation(t)]))(Tetris())),
                  ("RgbObservation(Tetris())", lambda: RgbObservation(Tetris()))):
    w = mk()
    _, info = w.reset(seed=1)
    st = {"info": info}

    def wstep():
        m = st["info"].get("action_mask")
        a = int(rng.choice(np.flatnonzero(m))) if m is not None else int(rng.integers(0, 8))
        _, _, term, _, st["info"] = w.step(a)
        if term:
            _, st["info"] = w.reset()

    print("%-42s median %.1f  min %.1f us" % ((label + ".step",) + med(wstep, warm=50)))
    w.close()
