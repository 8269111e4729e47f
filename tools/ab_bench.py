#!/usr/bin/env python3
"""Runs ON THE GPU BOX: A/B two builds of the engine (extra -D flags) inside ONE process, alternating the timed runs
(process-to-process variance on this pool is larger than most kernel changes).
usage: ab_bench.py [flagsA ...] -- [flagsB ...]     env: TG_FUSE (default 1 here), AB_BATCH"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
fa, fb = (args[:args.index("--")], args[args.index("--") + 1:]) if "--" in args else (args, [])
os.environ.setdefault("TG_FUSE", "1")
procs = []
for tag, flags in (("A", fa), ("B", fb)):
    root = os.environ.get("AB_SRC_" + tag, ROOT)  # (another source tree for this side, e.g. an older commit's csrc)
    src = os.path.join(root, "tetris_gymnasium_amd", "csrc", "tg_engine.hip")
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-DTG_FOR_EACH_SIZE(X)=X(10,20)", *flags, "-lhiprtc", "-shared", "-o", f"/tmp/libtg_{tag}.so", src]))
assert all(p.wait() == 0 for p in procs)
from tetris_gymnasium_amd import _native as N
from tetris_gymnasium_amd import Engine
B, K = int(os.environ.get("AB_BATCH", "65536")), 512
acts = np.random.default_rng(0).integers(0, 7 if os.environ.get('AB_RULESET') == 'functional' else 8, (K, B)).astype(np.int8)
import ctypes
for tag in "AB":  # older builds may lack entry points added later: bind only what both export
    lib = ctypes.CDLL(f"/tmp/libtg_{tag}.so")
    for name in list(N.SYMBOLS):
        if not hasattr(lib, name):
            del N.SYMBOLS[name]
eng = {}
for tag in ("A", "B", "A2", "B2"):  # two engines per build: buffer placement varies too
    N._lib = None
    N.LIB_PATH = f"/tmp/libtg_{tag[0]}.so"
    fnrules = os.environ.get("AB_RULESET", "numpy") == "functional"
    e = Engine(B, ruleset=(N.RULESET_FUNCTIONAL if fnrules else N.RULESET_NUMPY), obs_mode=(N.OBS_FN if fnrules else N.OBS_BOARD),
               autoreset=N.AUTORESET_SAME_STEP)
    e.reset(np.arange(1, B + 1, dtype=np.uint64))
    d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
    e.step_k_device(d, 64); e.sync()
    eng[tag] = (e, d)
res = {t: [] for t in eng}
for rep in range(6):
    for t, (e, d) in eng.items():
        e.timer_start(); e.step_k_device(d, K); res[t].append(e.timer_stop() / K * 1e3)
for t in eng:
    print(f"{t:3s} flags={fa if t[0] == 'A' else fb}: " + " ".join(f"{x:.2f}" for x in res[t]) + f"   min {min(res[t]):.2f} us/step")
