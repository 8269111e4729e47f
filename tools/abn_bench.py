#!/usr/bin/env python3
"""Runs ON THE GPU BOX: A/B/n of engine builds inside ONE process, alternating the timed runs (process-to-process and
box-to-box spread on this pool exceeds most kernel changes).  Every variant is a source tree (a copy of csrc under _ab/<tag>/
or the working tree) plus optional -D flags:

    tools/abn_bench.py tag[=srcdir][:-Dflag[:-Dflag...]][@ENV=value...] ...        env: AB_BATCH, AB_RULESET, AB_OBS, AB_STEPS, AB_ENGINES

@ENV=value pairs are set while the variant's engines are created (knobs tg_create reads, e.g. TG_SPLIT).

AB_TRAJ=1 (with TG_FUSE > 1): the timed launches go through tg_rollout with per-transition trajectory slots.
srcdir is a directory holding tg_engine.hip (default: tetris_gymnasium_amd/csrc).  Prints median / min us per step and a
checksum of the final state per variant (variants of the same rules must agree)."""
import ctypes, os, statistics, subprocess, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TG_FUSE", "1")
variants = []
envs = {}
for a in sys.argv[1:]:
    a, *kv = a.split("@")
    head, *flags = a.split(":")
    tag, _, src = head.partition("=")
    src = os.path.join(ROOT, src) if src else os.path.join(ROOT, "tetris_gymnasium_amd", "csrc")
    variants.append((tag, src, flags))
    envs[tag] = dict(x.split("=", 1) for x in kv)
procs = []
for tag, src, flags in variants:
    sched = [] if "-default-sched" in flags else ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]  # (variant flag -default-sched: the compiler's default strategy)
    flags = [f for f in flags if f != "-default-sched"]
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16", *sched,
                                   "-DTG_FOR_EACH_SIZE(X)=X(10,20)", f"-I{os.path.join(ROOT, 'tetris_gymnasium_amd', 'csrc')}", *flags,
                                   "-lhiprtc", "-shared", "-o", f"/tmp/libtg_{tag}.so", os.path.join(src, "tg_engine.hip")],
                                  stderr=subprocess.DEVNULL))
assert all(p.wait() == 0 for p in procs), "a variant failed to compile"
from tetris_gymnasium_amd import _native as N
from tetris_gymnasium_amd import Engine
B, K = int(os.environ.get("AB_BATCH", "65536")), int(os.environ.get("AB_STEPS", "512"))
fn = os.environ.get("AB_RULESET", "numpy") == "functional"
obs = {"board": N.OBS_BOARD, "dict": N.OBS_DICT, "fn": N.OBS_FN, "rgb": N.OBS_RGB}[os.environ.get("AB_OBS", "fn" if fn else "board")]
acts = np.random.default_rng(0).integers(0, 7 if fn else 8, (K, B)).astype(np.int8)
for tag, _, _ in variants:  # older builds may lack entry points added later: bind only what all export
    lib = ctypes.CDLL(f"/tmp/libtg_{tag}.so")
    for name in list(N.SYMBOLS):
        if not hasattr(lib, name):
            del N.SYMBOLS[name]
eng = {}
for tag, _, _ in variants:
    for rep in [""] + ["'" * i for i in range(1, int(os.environ.get("AB_ENGINES", "2")))]:  # several engines per build: buffer placement varies too
        N._lib = None
        N.LIB_PATH = f"/tmp/libtg_{tag}.so"
        os.environ.update(envs[tag])
        e = Engine(B, ruleset=(N.RULESET_FUNCTIONAL if fn else N.RULESET_NUMPY), obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP)
        e.reset(np.arange(1, B + 1, dtype=np.uint64))
        d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts)
        e.step_k_device(d, 64); e.sync()
        for k in envs[tag]:
            os.environ.pop(k, None)
        eng[tag + rep] = (e, d)
traj = os.environ.get("AB_TRAJ") == "1"
F = int(os.environ["TG_FUSE"])
if traj:
    e0 = next(iter(eng.values()))[0]
    ob = e0.lib.tg_obs_bytes_per_env(e0.cfg)
    tb = {t: (e.dev_alloc(F * B * ob), e.dev_alloc(F * B * 8)) for t, (e, d) in eng.items()}
def run(t, e, d):
    if not traj:
        e.step_k_device(d, K)
        return
    for k0 in range(0, K, F):
        e.rollout_device(d + k0 * B, min(F, K - k0), *tb[t])
res = {t: [] for t in eng}
for rep in range(7):
    for t, (e, d) in eng.items():
        e.timer_start(); run(t, e, d); res[t].append(e.timer_stop() / K * 1e3)
for t, (e, d) in eng.items():
    b, r, g = e.get_state()
    crc = zlib.crc32(b.tobytes()) ^ zlib.crc32(np.ascontiguousarray(r["bag"]).tobytes()) ^ zlib.crc32(g["state_lo"].tobytes())
    o = e.observation()
    for part in (o.values() if isinstance(o, dict) else [o]):  # (the last observation is part of the checksum)
        crc ^= zlib.crc32(np.ascontiguousarray(part).tobytes())
    print(f"{t:12s} median {statistics.median(res[t]):6.2f}  min {min(res[t]):6.2f} us/step   state crc {crc:08x}   " + " ".join(f"{x:.2f}" for x in res[t]))
for tag, _, _ in variants:
    ms = [statistics.median(res[t]) for t in eng if t.rstrip("'") == tag]
    print(f"{tag:12s} mean of the engines' medians {statistics.mean(ms):6.2f}   best {min(ms):6.2f}")
