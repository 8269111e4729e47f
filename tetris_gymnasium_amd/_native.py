"""ctypes binding of libtetris_gym_hip.so (the C ABI in include/tetris_gym_hip.h).

No PyTorch, no Triton, no JAX: plain ctypes over hand-written HIP.  There is no CPU fallback — if the
shared library is missing or cannot be loaded this module raises, it never degrades silently.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TG_LIB") or os.path.join(_PKG, "libtetris_gym_hip.so")  # TG_LIB: the test-only legacy build

ABI_VERSION = 6  # include/tetris_gym_hip.h TG_ABI_VERSION
RULESET_NUMPY, RULESET_FUNCTIONAL = 0, 1
OBS_NONE, OBS_BOARD, OBS_DICT, OBS_FN, OBS_RGB = 0, 1, 2, 3, 4
AUTORESET_NONE, AUTORESET_SAME_STEP, AUTORESET_NEXT_STEP = 0, 1, 2
ACTIONS_BASE, ACTIONS_GROUPED = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
(BUF_OBS, BUF_STEP_OUT, BUF_ACTIONS, BUF_BOARD, BUF_REC, BUF_RNG, BUF_LEGAL_MASK, BUF_FINAL_OBS,
 BUF_EPISODE, BUF_PLACEMENT_FEATS, BUF_FEATS, BUF_HOST_PACK) = range(12)
FLAG_EPISODE_STATS, FLAG_FINAL_OBS, FLAG_FEATURES, FLAG_CUSTOM_ACTIONS, FLAG_CUSTOM_REWARDS = 1, 2, 4, 8, 16
OUT_COMMITTED, OUT_ILLEGAL = 1, 2  # tg_step_out.flags
NONE = 0xFF
OK, EINVAL, EHIP, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4  # TG_E*
F_GAME_OVER, F_HAS_SWAPPED, F_PENDING_RESET = 1, 2, 4

ENV_REC = np.dtype(
    [
        ("piece", "u1"), ("rot", "u1"), ("x", "u1"), ("y", "u1"), ("val", "u1"), ("flags", "u1"),
        ("holder_piece", "u1"), ("holder_rot", "u1"), ("holder_val", "u1"), ("bag_index", "u1"),
        ("bag", "u1", (7,)), ("queue", "u1", (7,)), ("score", "<f4"), ("reserved", "<u4"),
    ]
)
RNG_REC = np.dtype(
    [
        ("state_lo", "<u8"), ("state_hi", "<u8"), ("inc_lo", "<u8"), ("inc_hi", "<u8"),
        ("uinteger", "<u4"), ("has_uint32", "<u4"), ("key", "<u4", (2,)),
    ]
)
EPISODE_REC = np.dtype([("ret", "<f4"), ("len", "<u4"), ("last_ret", "<f4"), ("last_len", "<u4")])
STEP_OUT = np.dtype([("reward", "<f4"), ("terminated", "u1"), ("lines", "u1"), ("flags", "u1"), ("reserved", "u1")])
assert ENV_REC.itemsize == 32 and RNG_REC.itemsize == 48 and EPISODE_REC.itemsize == 16 and STEP_OUT.itemsize == 8


class TgConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("padding", C.c_int32),
        ("queue_size", C.c_int32), ("gravity", C.c_int32), ("ruleset", C.c_int32), ("obs_mode", C.c_int32),
        ("autoreset", C.c_int32), ("action_set", C.c_int32), ("batch", C.c_int32), ("device", C.c_int32),
        ("env_offset", C.c_int32), ("grouped_terminate_on_illegal", C.c_int32), ("flags", C.c_int32),
        ("reserved", C.c_int32),
        # launch knobs (0 = default): transitions per launch of tg_step_k / tg_rollout, cadence and depth of the NUMPY bag look-ahead
        ("fuse", C.c_int32), ("rng_every", C.c_int32), ("ring_fill_p1", C.c_int32), ("streams", C.c_int32),
        # ABI 4: ActionsMapping field values in declaration order (FLAG_CUSTOM_ACTIONS) / RewardsMapping alife, clear_line,
        # game_over, invalid_action (FLAG_CUSTOM_REWARDS)
        ("action_ids", C.c_int32 * 8), ("rewards", C.c_double * 4),
    ]


# every symbol include/tetris_gym_hip.h declares: (restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "tg_abi_version": (_i, []),
    "tg_device_count": (_i, []),
    "tg_device_pci_bus_id": (_i, [_i, C.c_char_p, _i]),
    "tg_create": (_i, [C.POINTER(TgConfig), C.POINTER(_vp)]),
    "tg_destroy": (None, [_vp]),
    "tg_last_error": (C.c_char_p, [_vp]),
    "tg_reset": (_i, [_vp, _vp, _vp]),
    "tg_step": (_i, [_vp, _vp, _i]),
    "tg_step_k": (_i, [_vp, _vp, _i]),
    "tg_rollout": (_i, [_vp, _vp, _i, _vp, _vp]),
    "tg_grouped_rollout": (_i, [_vp, C.c_uint32, C.c_uint32, _i, _vp, _vp, _vp]),
    "tg_device_ptr": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    "tg_fetch": (_i, [_vp, _i, _vp, _sz]),
    "tg_host_pack_layout": (_i, [C.POINTER(TgConfig), C.POINTER(_sz)]),
    "tg_fetch_many": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz)]),
    "tg_step_fetch_many": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _i, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_sz),
                                C.POINTER(_sz)]),
    "tg_get_state": (_i, [_vp, _vp, _vp, _vp]),
    "tg_set_state": (_i, [_vp, _vp, _vp, _vp]),
    "tg_get_state_dev": (_i, [_vp, _vp, _vp, _vp]),
    "tg_set_state_dev": (_i, [_vp, _vp, _vp, _vp]),
    "tg_observe": (_i, [_vp]),
    "tg_grouped_enumerate": (_i, [_vp, _vp]),
    "tg_grouped_sample": (_i, [_vp, C.c_uint32, C.c_uint32]),
    "tg_features": (_i, [_vp]),
    "tg_observe_as": (_i, [_vp, _i, _vp, C.c_size_t]),
    "tg_features_as": (_i, [_vp, _vp, _vp]),
    "tg_sync": (_i, [_vp]),
    "tg_timer_start": (_i, [_vp]),
    "tg_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "tg_step_kernel_name": (_i, [_vp, _i, _i, _i, C.c_char_p, _sz]),
    "tg_seed_pcg64": (_i, [C.c_uint64, _vp]),
    "tg_obs_bytes_per_env": (_i, [C.POINTER(TgConfig)]),
    "tg_num_actions": (_i, [C.POINTER(TgConfig)]),
    "tg_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "tg_dev_free": (_i, [_vp, _vp]),
    "tg_host_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "tg_host_free": (_i, [_vp, _vp]),
    "tg_dev_upload": (_i, [_vp, _vp, _vp, _sz]),
    "tg_dev_download": (_i, [_vp, _vp, _vp, _sz]),
    "tg_dev_copy": (_i, [_vp, _vp, _vp, _sz]),
    "tg_stream_copy_gbps": (_i, [_vp, _sz, _i, C.POINTER(C.c_float)]),
    "tg_shape_mask": (_i, [_i, _i]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


HIP_RUNTIME = "system"  # which libamdhip64 the engine runs on: "system" (what it was built and tested against) or the path preloaded


def _share_torch_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so (same soname as /opt/rocm's): if this
    library loads the system copy first and torch is imported later, torch loads its bundled copy beside it and finds no GPU
    ("No HIP GPUs are available") — and zero-copy hand-over (`DeviceArray`) needs both sides in one runtime anyway.

    The engine is built and tested against the SYSTEM runtime, so that is the default and nothing is preloaded.  The bundled
    runtime is taken only
      * when torch has ALREADY been imported by this process (`"torch" in sys.modules`: its runtime is loaded anyway, the
        dynamic linker resolves our libamdhip64.so.N to it if the sonames match — preloading just makes that explicit), or
      * on request: TG_HIP_RUNTIME=torch (engine first, torch later in the same process).
    Either way only a bundled library whose SONAME equals the one this engine links against is accepted (two different
    runtimes in one process is the failure this exists to avoid); TG_HIP_RUNTIME=system never preloads.  The choice is
    recorded in `_native.HIP_RUNTIME` (and printed with TG_VERBOSE=1)."""
    global HIP_RUNTIME
    import sys

    mode = os.environ.get("TG_HIP_RUNTIME", "auto")
    if mode == "system" or (mode != "torch" and "torch" not in sys.modules):
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if not os.path.exists(cand):
            return
        want, have = _needed_hip_soname(LIB_PATH), _soname(cand)
        if want and have and want != have:
            if os.environ.get("TG_VERBOSE"):
                print(f"tetris_gymnasium_amd: torch bundles {have}, the engine links {want}: keeping the system runtime",
                      file=sys.stderr)
            return
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        HIP_RUNTIME = cand
    except Exception:  # a broken / non-ROCm torch installation must not keep the engine from loading
        pass


def _elf_strings(path, tag):
    """DT_SONAME (14) / DT_NEEDED (1) strings of an ELF64 shared object, read without loading it."""
    import struct

    with open(path, "rb") as f:
        d = f.read()
    if d[:4] != b"\x7fELF" or d[4] != 2:
        return []
    shoff, shentsize, shnum = struct.unpack_from("<Q", d, 0x28)[0], *struct.unpack_from("<HH", d, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize) for i in range(shnum)]
    out = []
    for s in secs:
        if s[1] != 6:  # SHT_DYNAMIC
            continue
        strtab = secs[s[6]]
        for o in range(s[4], s[4] + s[5], 16):
            t, v = struct.unpack_from("<qQ", d, o)
            if t == 0:
                break
            if t == tag:
                e = d.index(b"\0", strtab[4] + v)
                out.append(d[strtab[4] + v:e].decode())
    return out


def _soname(path):
    try:
        r = _elf_strings(os.path.realpath(path), 14)
        return r[0] if r else None
    except Exception:
        return None


def _needed_hip_soname(path):
    try:
        return next((n for n in _elf_strings(path, 1) if n.startswith("libamdhip64")), None)
    except Exception:
        return None


def load() -> C.CDLL:
    """Load the HIP engine.  Raises NativeLibraryError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C tetris_gymnasium_amd/csrc` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_torch_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    if hasattr(lib, "tgo_create") and os.environ.get("TG_LIB_IS_CPU_CHECKER") != "1":
        # (TG_LIB pointing at the test infrastructure's CPU implementation of this ABI: a bisect tool for developers, never a
        # fallback — it is taken only when asked for twice)
        raise NativeLibraryError(f"{LIB_PATH} is the CPU checker's implementation of the C ABI, not the HIP engine: set "
                                 f"TG_LIB_IS_CPU_CHECKER=1 as well to drive the host side against it (tests only)")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.tg_abi_version() != ABI_VERSION:
        raise NativeLibraryError("libtetris_gym_hip.so ABI version mismatch")
    _lib = lib
    if os.environ.get("TG_VERBOSE"):
        import sys

        print(f"tetris_gymnasium_amd: HIP runtime = {HIP_RUNTIME}", file=sys.stderr)
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
