"""`Engine`: the batched device-resident Tetris state behind the C ABI (one handle = one GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


class EngineError(RuntimeError):
    pass


# ---- device-resident views ------------------------------------------------------------------------------------------
class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int), ("device_id", C.c_int)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DLDeleter = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DLDeleter)]
_KDLROCM = 10  # DLDeviceType::kDLROCM
_dl_live: dict = {}  # address of a handed-out DLManagedTensor → everything that must outlive the capsule


@_DLDeleter
def _dl_deleter(mt):
    _dl_live.pop(C.addressof(mt.contents), None)


_PyCapsule_Destructor = C.CFUNCTYPE(None, C.c_void_p)
_PyCapsule_New = C.pythonapi.PyCapsule_New
_PyCapsule_New.restype = C.py_object
_PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, _PyCapsule_Destructor]
_PyCapsule_IsValid = C.pythonapi.PyCapsule_IsValid
_PyCapsule_IsValid.restype = C.c_int
_PyCapsule_IsValid.argtypes = [C.c_void_p, C.c_char_p]
_PyCapsule_GetPointer = C.pythonapi.PyCapsule_GetPointer
_PyCapsule_GetPointer.restype = C.c_void_p
_PyCapsule_GetPointer.argtypes = [C.c_void_p, C.c_char_p]


@_PyCapsule_Destructor
def _dl_capsule_destructor(capsule):
    """DLPack protocol: a capsule that dies still named "dltensor" was never consumed (the consumer renames it to
    "used_dltensor" when it takes ownership) — the producer's deleter must then run, or the view, its pooled device
    buffer and the engine reference it pins stay alive for ever."""
    if _PyCapsule_IsValid(capsule, b"dltensor"):
        _dl_live.pop(_PyCapsule_GetPointer(capsule, b"dltensor"), None)



class DeviceArray:
    """A typed, shaped view of device memory that other GPU libraries take without a copy: `__cuda_array_interface__`
    (version 3; `torch.as_tensor(view, device="cuda")`, cupy, numba) and `__dlpack__` (`torch.from_dlpack(view)`; device type
    kDLROCM).  `numpy()` / `np.asarray(view)` download it.  Views of engine buffers (Engine.device_array) alias memory the
    next transition overwrites — copy on the consumer's side to keep a value; an `owner` (a pooled device buffer) makes the
    view own its bytes.  Exporting waits for the engine's stream first (`sync=True`), so the consumer reads finished data."""

    def __init__(self, engine, ptr, shape, dtype, strides=None, owner=None, sync=True, readonly=False, refresh=None):
        self.engine, self.ptr, self.shape, self.dtype = engine, int(ptr), tuple(int(x) for x in shape), np.dtype(dtype)
        self.strides = None if strides is None else tuple(int(x) for x in strides)
        self._owner, self._sync, self._readonly = owner, sync, readonly
        self._refresh = refresh  # engine buffer that is brought up to date on demand (TG_BUF_BOARD / TG_BUF_RNG / the legal mask)

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def _ready(self):
        if not getattr(self.engine, "h", None) or (self._owner is not None and not self._owner.ptr):
            raise EngineError("this DeviceArray's engine has been closed: its device memory is gone")
        if self._refresh is not None:  # e.g. the uint8 boards of a handle that keeps its board as nibbles: a view held across
            self.engine.device_ptr(self._refresh)  # transitions would otherwise show the boards of the moment it was created
        if self._sync:
            self.engine.sync()

    @property
    def __cuda_array_interface__(self):
        self._ready()
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, self._readonly), "version": 3,
                "strides": self.strides, "descr": [("", self.dtype.str)], "stream": None}

    def __dlpack_device__(self):
        return (_KDLROCM, int(self.engine.cfg.device))

    def __dlpack__(self, stream=None, **_):
        # (`stream`: the consumer's stream.  Export waits for the engine's stream on the host first — `sync=True` views — so
        # the data is complete whatever stream reads it; a view created with sync=False leaves ordering to the caller.)
        self._ready()
        nd = len(self.shape)
        shape = (C.c_int64 * nd)(*self.shape)
        item = self.dtype.itemsize
        strides = None if self.strides is None else (C.c_int64 * nd)(*[st // item for st in self.strides])
        code = {"u": 1, "i": 0, "f": 2, "b": 6}[self.dtype.kind]
        mt = _DLManagedTensor()
        mt.dl_tensor = _DLTensor(C.c_void_p(self.ptr), _DLDevice(_KDLROCM, int(self.engine.cfg.device)), nd,
                                 _DLDataType(code, item * 8, 1), shape,
                                 C.cast(strides, C.POINTER(C.c_int64)) if strides is not None else None, 0)
        mt.manager_ctx = None
        mt.deleter = _dl_deleter
        _dl_live[C.addressof(mt)] = (mt, shape, strides, self)  # the view (and its owner) lives as long as the consumer's tensor
        return _PyCapsule_New(C.addressof(mt), b"dltensor", _dl_capsule_destructor)

    def numpy(self):
        """Download (waits for the engine's stream)."""
        if self.strides is None:
            out = np.empty(self.shape, self.dtype)
            self.engine.dev_download(self.ptr, out)
            return out
        # strided view of a record array: fetch the enclosing bytes once, then view
        span = sum((n - 1) * st for n, st in zip(self.shape, self.strides)) + self.dtype.itemsize
        raw = np.empty(span, np.uint8)
        self.engine.dev_download(self.ptr, raw)
        return np.lib.stride_tricks.as_strided(raw[: self.dtype.itemsize].view(self.dtype), self.shape, self.strides).copy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.engine.cfg.device}, ptr=0x{self.ptr:x})"


class DeviceBuffer:
    """A device allocation from the engine's free list (hipMalloc / hipFree cost ~100 us and a stream sync each: buffers of
    the sizes a loop keeps asking for are recycled).  Returns itself to the list when the last view dies."""

    def __init__(self, engine, nbytes):
        self.engine, self.nbytes = engine, int(nbytes)
        free = engine._pool.setdefault(self.nbytes, [])
        self.ptr = free.pop() if free else engine.dev_alloc(self.nbytes)
        engine._live_bufs.add(self)  # (weakly: Engine.close frees what is still out and zeroes `ptr`, so late users fail loudly)

    def __del__(self):
        try:
            if self.engine.h and self.ptr:
                self.engine._pool.setdefault(self.nbytes, []).append(self.ptr)
        except Exception:
            pass


class _PinnedBlock:
    """A page-locked host allocation (tg_host_alloc) that lives as long as any NumPy view of it: arrays made from it keep the
    block as their base object, and the block's finalizer returns the memory (tg_host_free with a NULL handle needs no
    engine).  Engine.close() therefore never pulls memory from under an array the caller still holds — gymnasium's
    `copy=False` arrays stay valid after `envs.close()`, so do these."""

    def __init__(self, lib, handle, nbytes):
        import weakref

        self.nbytes = max(int(nbytes), 1)
        p = C.c_void_p()
        rc = lib.tg_host_alloc(handle, self.nbytes, C.byref(p))
        if rc != 0:
            raise EngineError(f"tg_host_alloc failed ({rc}): {lib.tg_last_error(handle).decode()}")
        self.ptr = p.value
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}
        weakref.finalize(self, lib.tg_host_free, None, C.c_void_p(self.ptr))

    def array(self, nbytes=None):
        a = np.asarray(self)  # (base = this block)
        return a if nbytes is None else a[:nbytes]


def mapping_values(m, n, cast):
    """The field values of a mappings dataclass in declaration order (or a plain sequence of them) as a tuple; None stays None."""
    if m is None:
        return None
    import dataclasses

    vals = [getattr(m, f.name) for f in dataclasses.fields(m)] if dataclasses.is_dataclass(m) else list(m)
    if len(vals) != n:
        raise ValueError(f"mapping with {len(vals)} fields: the engine implements the reference's {n} (mappings/*.py)")
    return tuple(cast(v) for v in vals)


class Engine:
    """Owns `batch` environments on one MI355X.  All arrays in/out are NumPy; state stays on the device.

    ruleset NUMPY mirrors tetris_gymnasium/envs/tetris.py (Tetris.step / reset), FUNCTIONAL mirrors
    tetris_gymnasium/envs/tetris_fn.py (step / reset).
    """

    def __init__(self, batch, width=10, height=20, *, ruleset=N.RULESET_NUMPY, gravity=True, queue_size=None,
                 obs_mode=None, autoreset=N.AUTORESET_NONE, device=0, env_offset=0, padding=4, episode_stats=False,
                 final_obs=False, grouped=False, terminate_on_illegal=True, features=False, fuse=0, rng_every=0,
                 ring_fill=None, actions_mapping=None, rewards_mapping=None, streams=0):
        """fuse / rng_every / ring_fill: launch knobs of the step kernel (tg_config; 0 / None = the defaults 16 / 3 / 3).
        Results never depend on them.
        actions_mapping / rewards_mapping (NUMPY rule-set): an ActionsMapping / RewardsMapping instance, or the field values in
        declaration order (8 ints / 4 numbers) — per-handle data of the step kernel (reference envs/tetris.py:83-84, 189-190).
        None, or a mapping equal to the stock one, = the stock tables."""
        import weakref

        self.lib = N.load()
        self._pool = {}      # free device buffers by size (DeviceBuffer)
        self._live_bufs = weakref.WeakSet()  # DeviceBuffers handed out and not yet returned
        self._pinned, self._pin_slots = {}, 0  # page-locked host buffers (use_pinned_observations): views of _PinnedBlocks
        if queue_size is None:
            queue_size = 4 if ruleset == N.RULESET_NUMPY else 7
        if obs_mode is None:
            obs_mode = N.OBS_DICT if ruleset == N.RULESET_NUMPY else N.OBS_FN
        cfg = N.TgConfig()
        cfg.struct_size = C.sizeof(N.TgConfig)
        cfg.width, cfg.height, cfg.padding, cfg.queue_size = width, height, padding, queue_size
        cfg.gravity, cfg.ruleset, cfg.obs_mode, cfg.autoreset = int(bool(gravity)), ruleset, obs_mode, autoreset
        cfg.action_set = N.ACTIONS_GROUPED if grouped else N.ACTIONS_BASE
        cfg.batch, cfg.device, cfg.env_offset = batch, device, env_offset
        cfg.grouped_terminate_on_illegal = int(bool(terminate_on_illegal))
        cfg.flags = ((N.FLAG_EPISODE_STATS if episode_stats else 0) | (N.FLAG_FINAL_OBS if final_obs else 0)
                     | (N.FLAG_FEATURES if features else 0))
        cfg.fuse, cfg.rng_every = int(fuse), int(rng_every)
        cfg.ring_fill_p1 = 0 if ring_fill is None else int(ring_fill) + 1
        cfg.streams = int(streams)  # (> 1: env shards on several streams — off by default, measured slower)
        ids, rws = mapping_values(actions_mapping, 8, int), mapping_values(rewards_mapping, 4, float)
        self.action_ids = tuple(range(8)) if ids is None else ids
        self.rewards = (1.0, 1.0, 0.0, -0.1) if rws is None else rws
        if ids is not None and ids != tuple(range(8)):
            cfg.flags |= N.FLAG_CUSTOM_ACTIONS
            cfg.action_ids[:] = ids
        if rws is not None and rws != (1.0, 1.0, 0.0, -0.1):
            cfg.flags |= N.FLAG_CUSTOM_REWARDS
            cfg.rewards[:] = rws
        self.cfg = cfg
        self.B, self.W, self.H, self.P, self.Q = batch, width, height, padding, queue_size
        self.Wp, self.Hp = width + 2 * padding, height + padding
        self.ruleset, self.obs_mode, self.autoreset = ruleset, obs_mode, autoreset
        self.n_actions = self.lib.tg_num_actions(C.byref(cfg))
        h = C.c_void_p()
        rc = self.lib.tg_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(f"tg_create failed ({rc}): {self.lib.tg_last_error(None).decode()}")
        self.h = h

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            for ptrs in self._pool.values():
                for p in ptrs:
                    self.lib.tg_dev_free(self.h, C.c_void_p(p))
            self._pool = {}
            for b in list(self._live_bufs):  # buffers still referenced by States / DeviceArrays: freed now, marked dead
                if b.ptr:
                    self.lib.tg_dev_free(self.h, C.c_void_p(b.ptr))
                    b.ptr = 0
            self._pinned = {}  # (the page-locked blocks free themselves when the last array viewing them dies)
            self._pin_state = self._stage_blk = self._sfo = None
            self.lib.tg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.tg_last_error(self.h).decode()}")

    # -- transitions ---------------------------------------------------------------------------
    def reset(self, seeds=None, mask=None):
        s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if s is not None and s.shape != (self.B,):
            raise ValueError("seeds must have shape (batch,)")
        if m is not None and m.shape != (self.B,):
            raise ValueError("mask must have shape (batch,)")
        self._chk(self.lib.tg_reset(self.h, N.ptr(s), N.ptr(m)), "tg_reset")

    def step_async(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.B,):
            raise ValueError("actions must have shape (batch,)")
        self._chk(self.lib.tg_step(self.h, N.ptr(a), N.MEM_HOST), "tg_step")

    def step_device(self, actions_dev_ptr):
        """actions: raw device pointer to int32[B] (e.g. from another library's tensor)."""
        self._chk(self.lib.tg_step(self.h, C.c_void_p(actions_dev_ptr), N.MEM_DEVICE), "tg_step")

    def step_k_device(self, actions_dev_ptr, k):
        self._chk(self.lib.tg_step_k(self.h, C.c_void_p(actions_dev_ptr), int(k)), "tg_step_k")

    def rollout_device(self, actions_dev_ptr, k, obs_traj_dev_ptr=None, out_traj_dev_ptr=None):
        """tg_rollout between DEVICE buffers of the caller (asynchronous): actions int8[k][B]; the observation of transition j
        goes to obs_traj + j * B * obs_bytes, its outputs to out_traj[j][B] (either may be None)."""
        self._chk(self.lib.tg_rollout(self.h, C.c_void_p(actions_dev_ptr), int(k),
                                      C.c_void_p(obs_traj_dev_ptr) if obs_traj_dev_ptr else None,
                                      C.c_void_p(out_traj_dev_ptr) if out_traj_dev_ptr else None), "tg_rollout")

    def rollout(self, actions, want_obs=True):
        """K transitions with the given actions int8[K][B] in one call (tg_rollout: the default kernel runs up to 16 per
        launch with the state resident on chip).  Returns (obs[K][B]..., reward f32[K][B], terminated u8[K][B],
        lines u8[K][B]) — what K calls of step() + observation() would have returned."""
        a = np.asarray(actions)
        if a.ndim != 2 or a.shape[1] != self.B:
            raise ValueError("actions must have shape (K, batch)")
        if self.ruleset == N.RULESET_NUMPY and ((a < 0) | (a >= self.n_actions)).any():
            # envs/tetris.py:215-217 asserts action_space.contains(action); device-resident actions (tg_step_k / tg_rollout
            # with device pointers) are NOT range-checked: the kernel takes `action & 7` of the base action set
            raise ValueError(f"action outside Discrete({self.n_actions})")
        a = np.ascontiguousarray(a, dtype=np.int8)
        K = a.shape[0]
        per_env = self.lib.tg_obs_bytes_per_env(C.byref(self.cfg))
        b_a = DeviceBuffer(self, a.nbytes)  # (scratch from the engine's free list: no hipMalloc / hipFree + sync per call)
        b_o = DeviceBuffer(self, K * self.B * per_env) if (per_env and want_obs) else None
        b_r = DeviceBuffer(self, K * self.B * N.STEP_OUT.itemsize)
        d_a, d_o, d_r = b_a.ptr, (b_o.ptr if b_o else None), b_r.ptr
        try:
            self.dev_upload(d_a, a)
            self._chk(self.lib.tg_rollout(self.h, C.c_void_p(d_a), K, C.c_void_p(d_o) if d_o else None, C.c_void_p(d_r)),
                      "tg_rollout")
            out = np.empty((K, self.B), N.STEP_OUT)
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(out), C.c_void_p(d_r), out.nbytes), "tg_dev_download")
            obs = None
            if d_o:
                raw = np.empty(K * self.B * per_env, np.uint8)
                self._chk(self.lib.tg_dev_download(self.h, N.ptr(raw), C.c_void_p(d_o), raw.nbytes), "tg_dev_download")
                if self.obs_mode == N.OBS_BOARD:
                    obs = raw.reshape(K, self.B, self.H, self.W)
                elif self.obs_mode == N.OBS_FN:
                    obs = raw.view(np.int8).reshape(K, self.B, self.H, self.W)
                elif self.obs_mode == N.OBS_RGB:
                    obs = raw.reshape(K, self.B, self.Hp, self.Wp + self.P * max(1, self.Q), 3)
                else:
                    obs = raw.reshape(K, self.B * per_env)  # DICT: the engine's flat layout per transition
        finally:
            del b_a, b_o, b_r
        return (obs, np.ascontiguousarray(out["reward"]), np.ascontiguousarray(out["terminated"]),
                np.ascontiguousarray(out["lines"]))

    def step(self, actions):
        """One transition; returns (reward f32[B], terminated u8[B], lines u8[B])."""
        self.step_async(actions)
        return self.fetch_outputs()

    def step_fetch_out(self, actions):
        """One transition from host actions and its tg_step_out records, in ONE call into the C ABI (tg_step_fetch_many: upload,
        launch, download, one wait) through page-locked buffers — what the single-env wrappers do per step."""
        fs = getattr(self, "_sfo", None)
        if fs is None:
            act = self.pinned_bytes(4 * self.B).view(np.int32)
            out = self.pinned_bytes(self.B * N.STEP_OUT.itemsize).view(N.STEP_OUT)
            fs = self._sfo = (act, out, (C.c_void_p * 1)(self.h.value), (C.c_void_p * 1)(act.ctypes.data), (C.c_int * 1)(N.BUF_STEP_OUT),
                              (C.c_void_p * 1)(out.ctypes.data), (C.c_size_t * 1)(out.nbytes))
        act, out, hs, ap, which, dst, nb = fs
        act[:] = np.asarray(actions, np.int32).reshape(self.B)
        self._chk(self.lib.tg_step_fetch_many(1, hs, ap, N.MEM_HOST, 1, which, dst, nb, None), "tg_step_fetch_many")
        return out.copy()

    def fetch_step_out(self):
        """TG_BUF_STEP_OUT of the last transition as a tg_step_out record array (reward f32, terminated, lines, flags)."""
        return self._fetch(N.BUF_STEP_OUT, self._host_array((self.B,), N.STEP_OUT, "out"))

    def host_pack(self):
        """TG_BUF_HOST_PACK of the last transition: what a gymnasium-contract step() returns beside the observation, in its final
        dtypes — {"rewards": f64[B] (the Python value: int score + the RewardsMapping's floats), "episode_returns": f64[B],
        "lines_cleared": i32[B], "episode_lengths": i32[B], "terminated": bool[B]}."""
        off = (C.c_size_t * 6)()
        self._chk(self.lib.tg_host_pack_layout(C.byref(self.cfg), off), "tg_host_pack_layout")
        raw = np.empty(int(off[5]), np.uint8)
        self._chk(self.lib.tg_fetch(self.h, N.BUF_HOST_PACK, N.ptr(raw), raw.nbytes), "tg_fetch(host_pack)")
        B = self.B
        return {"rewards": raw[off[0]:off[0] + 8 * B].view(np.float64), "episode_returns": raw[off[1]:off[1] + 8 * B].view(np.float64),
                "lines_cleared": raw[off[2]:off[2] + 4 * B].view(np.int32), "episode_lengths": raw[off[3]:off[3] + 4 * B].view(np.int32),
                "terminated": raw[off[4]:off[4] + B].view(np.bool_)}

    def fetch_outputs(self):
        out = self._fetch(N.BUF_STEP_OUT, self._host_array((self.B,), N.STEP_OUT, "out"))
        return (np.ascontiguousarray(out["reward"]), np.ascontiguousarray(out["terminated"]),
                np.ascontiguousarray(out["lines"]))

    # -- observations --------------------------------------------------------------------------
    _SMALL = 1 << 16

    def _stage(self, nbytes):
        """uint8[nbytes] of one page-locked block kept for SMALL host transfers: a copy to or from pageable memory is staged and
        waited for inside the runtime (≈20 us per call on a one-env handle, tools/single_env_latency.py); through this block it is
        a plain DMA plus a memcpy of a few KB.  Every user of the block waits for its copy before it returns."""
        blk = getattr(self, "_stage_blk", None)
        if blk is None or blk.nbytes < nbytes:
            blk = self._stage_blk = self.pinned_bytes(max(int(nbytes), self._SMALL))
        return blk[:nbytes]

    def _fetch(self, which, arr):
        n = arr.nbytes
        if 0 < n <= self._SMALL and not self._pin_slots and arr.flags.c_contiguous:
            st = self._stage(n)
            self._chk(self.lib.tg_fetch(self.h, which, N.ptr(st), n), f"tg_fetch({which})")  # (waits for the copy)
            arr.reshape(-1).view(np.uint8)[:] = st
            return arr
        self._chk(self.lib.tg_fetch(self.h, which, N.ptr(arr), n), f"tg_fetch({which})")
        return arr

    def use_pinned_observations(self, slots=2):
        """observation() / fetch_outputs() return views of `slots` rotating page-locked host buffers instead of fresh NumPy
        arrays: the device-to-host copy is one DMA at PCIe rate and nothing is allocated per call.  A returned array is
        overwritten `slots` calls later (gymnasium's `copy=False`); the buffers live until close()."""
        self._pin_slots = max(2, int(slots))

    def pinned_bytes(self, nbytes):
        """uint8[nbytes] in page-locked host memory that frees itself when the last view of it dies (_PinnedBlock)"""
        return _PinnedBlock(self.lib, self.h, nbytes).array(nbytes)

    def _host_array(self, shape, dtype, tag):
        """a host array for a fetch: fresh (default) or the next page-locked slot of its kind"""
        if not getattr(self, "_pin_slots", 0):
            return np.empty(shape, dtype)
        ring = self._pinned.setdefault((tag, tuple(shape), np.dtype(dtype).str), [])
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if len(ring) < self._pin_slots:
            ring.append(self.pinned_bytes(nbytes).view(dtype).reshape(shape))
            return ring[-1]
        ring.append(ring.pop(0))
        return ring[-1]

    def observation(self, which=N.BUF_OBS):
        """The observation buffer of the engine's obs_mode as NumPy array(s)."""
        if self.obs_mode == N.OBS_BOARD:
            return self._fetch(which, self._host_array((self.B, self.H, self.W), np.uint8, which))
        if self.obs_mode == N.OBS_FN:
            return self._fetch(which, self._host_array((self.B, self.H, self.W), np.int8, which))
        if self.obs_mode == N.OBS_DICT:
            cells, pp, B = self.Hp * self.Wp, self.P * self.P, self.B
            flat = self._fetch(which, self._host_array((B * (2 * cells + pp + pp * self.Q),), np.uint8, which))
            o = 0
            board = flat[o:o + B * cells].reshape(B, self.Hp, self.Wp); o += B * cells
            mask = flat[o:o + B * cells].reshape(B, self.Hp, self.Wp); o += B * cells
            holder = flat[o:o + B * pp].reshape(B, self.P, self.P); o += B * pp
            queue = flat[o:o + B * pp * self.Q].reshape(B, self.P, self.P * self.Q)
            return {"board": board, "active_tetromino_mask": mask, "holder": holder, "queue": queue}
        if self.obs_mode == N.OBS_RGB:
            return self._fetch(which, self._host_array((self.B, self.Hp, self.Wp + self.P * max(1, self.Q), 3), np.uint8, which))
        raise EngineError("engine was created with obs_mode NONE")

    def device_array(self, which=N.BUF_OBS):
        """Zero-copy view(s) of an engine-owned buffer for other GPU libraries (DeviceArray: `__cuda_array_interface__` +
        DLPack): the observation in the shape(s) `observation()` returns (a dict of four views for the dict surface), the
        last transition's outputs as {"reward" f32[B], "terminated" u8[B], "lines" u8[B]} (strided views of tg_step_out),
        the legal mask, the boards, the actions buffer.  What a trainer does with `torch.Tensor(next_obs).to(device)`
        (examples/train_ppo.py:240-245) without the trip through host memory."""
        ptr, nbytes = self.device_ptr(which)
        B = self.B
        if which in (N.BUF_OBS, N.BUF_FINAL_OBS):
            if self.obs_mode == N.OBS_BOARD:
                return DeviceArray(self, ptr, (B, self.H, self.W), np.uint8)
            if self.obs_mode == N.OBS_FN:
                return DeviceArray(self, ptr, (B, self.H, self.W), np.int8)
            if self.obs_mode == N.OBS_RGB:
                return DeviceArray(self, ptr, (B, self.Hp, self.Wp + self.P * max(1, self.Q), 3), np.uint8)
            if self.obs_mode == N.OBS_DICT:
                cells, pp = self.Hp * self.Wp, self.P * self.P
                return {"board": DeviceArray(self, ptr, (B, self.Hp, self.Wp), np.uint8),
                        "active_tetromino_mask": DeviceArray(self, ptr + B * cells, (B, self.Hp, self.Wp), np.uint8),
                        "holder": DeviceArray(self, ptr + 2 * B * cells, (B, self.P, self.P), np.uint8),
                        "queue": DeviceArray(self, ptr + 2 * B * cells + B * pp, (B, self.P, self.P * self.Q), np.uint8)}
            raise EngineError("engine was created with obs_mode NONE")
        if which == N.BUF_STEP_OUT:
            st = N.STEP_OUT.itemsize
            return {"reward": DeviceArray(self, ptr, (B,), np.float32, strides=(st,)),
                    "terminated": DeviceArray(self, ptr + 4, (B,), np.uint8, strides=(st,)),
                    "lines": DeviceArray(self, ptr + 5, (B,), np.uint8, strides=(st,))}
        if which == N.BUF_LEGAL_MASK:
            return DeviceArray(self, ptr, (B, 4 * self.W), np.uint8)
        if which == N.BUF_BOARD:  # (materialised from the packed board on demand on handles of >= 20 Mi cells: re-done at every export)
            return DeviceArray(self, ptr, (B, self.H, self.W), np.uint8, refresh=N.BUF_BOARD)
        if which == N.BUF_ACTIONS:
            return DeviceArray(self, ptr, (B,), np.int32)
        if which == N.BUF_FEATS:
            return DeviceArray(self, ptr, (B, self.W + 3), np.uint8)
        if which == N.BUF_PLACEMENT_FEATS:
            return DeviceArray(self, ptr, (B, 4 * self.W, self.W + 3), np.uint8)
        return DeviceArray(self, ptr, (nbytes,), np.uint8)

    def dev_copy(self, dst_ptr, src_ptr, nbytes):
        """Device → device, asynchronous on the engine's stream."""
        self._chk(self.lib.tg_dev_copy(self.h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes), "tg_dev_copy")

    def get_state_device(self):
        """Snapshot of (board, rec, rng) in pooled device buffers (asynchronous; envs/tetris.py:681-708 without the host)."""
        bufs = (DeviceBuffer(self, self.B * self.H * self.W), DeviceBuffer(self, self.B * N.ENV_REC.itemsize),
                DeviceBuffer(self, self.B * N.RNG_REC.itemsize))
        self._chk(self.lib.tg_get_state_dev(self.h, C.c_void_p(bufs[0].ptr), C.c_void_p(bufs[1].ptr), C.c_void_p(bufs[2].ptr)),
                  "tg_get_state_dev")
        for role, b in zip(("board", "rec", "rng"), bufs):
            b._state_of = (role, self.B, self.W, self.H, self.ruleset)  # (what set_state_device accepts in that slot without `unchecked`)
        return bufs

    def set_state_device(self, board=None, rec=None, rng=None, unchecked=False):
        """Restore from device buffers — the counterpart of get_state_device.  By default only snapshots that get_state_device of
        an engine of the same shape produced are accepted; unchecked=True takes raw device pointers / foreign buffers.  Either
        way tg_set_state_dev range-checks the records (and, where cell values are restricted, the boards) ON THE DEVICE: what
        tg_set_state would refuse is clamped so that nothing indexes out of range, and the next call that waits for the engine
        (sync / fetch / get_state / step with host results) raises EngineError once."""
        for role, b in zip(("board", "rec", "rng"), (board, rec, rng)):
            if b is None or unchecked:
                continue
            # (tagged per role: the rec buffer passed as `board` would make tg_set_state_dev read B*H*W bytes out of a B*32-byte allocation)
            if not isinstance(b, DeviceBuffer) or getattr(b, "_state_of", None) != (role, self.B, self.W, self.H, self.ruleset):
                raise EngineError(f"set_state_device({role}=...) takes the `{role}` DeviceBuffer get_state_device returned for an engine of "
                                  "this shape (device records are not validated); pass unchecked=True for anything else")
        p = [None if b is None else C.c_void_p(b.ptr if isinstance(b, DeviceBuffer) else int(b)) for b in (board, rec, rng)]
        self._chk(self.lib.tg_set_state_dev(self.h, *p), "tg_set_state_dev")

    # -- grouped action surface (wrappers/grouped.py) ---------------------------------------------
    def legal_mask(self):
        """u8 [B, 4*W]: 1 = legal placement (column*4 + rotations) for the current active pieces."""
        self._chk(self.lib.tg_grouped_enumerate(self.h, None), "tg_grouped_enumerate")
        return self._fetch(N.BUF_LEGAL_MASK, np.empty((self.B, 4 * self.W), np.uint8))

    def placement_boards(self):
        """(legal mask u8[B][4W], boards u8[B][4W][H+p][W+2p]): the observation of GroupedActionsObservations
        (wrappers/grouped.py:124-207) for the current state, built on the device."""
        n = self.B * 4 * self.W * self.Hp * self.Wp
        buf = DeviceBuffer(self, n)
        self._chk(self.lib.tg_grouped_enumerate(self.h, C.c_void_p(buf.ptr)), "tg_grouped_enumerate(boards)")
        boards = np.empty((self.B, 4 * self.W, self.Hp, self.Wp), np.uint8)
        self.dev_download(buf.ptr, boards)
        del buf
        return self._fetch(N.BUF_LEGAL_MASK, np.empty((self.B, 4 * self.W), np.uint8)), boards

    def placement_boards_device(self, boards_dev_ptr):
        """Same, into a caller-owned device buffer (asynchronous on the engine's stream)."""
        self._chk(self.lib.tg_grouped_enumerate(self.h, C.c_void_p(boards_dev_ptr)), "tg_grouped_enumerate(boards)")

    def sample_legal_actions(self, seed=0, counter=0, fetch=True):
        """Draw one legal action per env on the device into the actions buffer (uniform over the mask)."""
        self._chk(self.lib.tg_grouped_sample(self.h, int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF), "tg_grouped_sample")
        if fetch:
            return self._fetch(N.BUF_ACTIONS, np.empty(self.B, np.int32))

    def grouped_rollout(self, K, seed=0, counter0=0, fetch=True):
        """K transitions of the uniform-random-legal policy drawn on the device (tg_grouped_rollout).  With fetch=True
        returns (actions i8[K][B], reward f32[K][B], terminated u8[K][B], lines u8[K][B]); fetch=False only enqueues."""
        if not fetch:
            self._chk(self.lib.tg_grouped_rollout(self.h, int(seed) & 0xFFFFFFFF, int(counter0) & 0xFFFFFFFF, int(K),
                                                  None, None, None), "tg_grouped_rollout")
            return None
        b_a, b_r = DeviceBuffer(self, K * self.B), DeviceBuffer(self, K * self.B * N.STEP_OUT.itemsize)
        d_a, d_r = b_a.ptr, b_r.ptr
        try:
            self._chk(self.lib.tg_grouped_rollout(self.h, int(seed) & 0xFFFFFFFF, int(counter0) & 0xFFFFFFFF, int(K),
                                                  C.c_void_p(d_a), None, C.c_void_p(d_r)), "tg_grouped_rollout")
            acts = np.empty((K, self.B), np.int8)
            out = np.empty((K, self.B), N.STEP_OUT)
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(acts), C.c_void_p(d_a), acts.nbytes), "tg_dev_download")
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(out), C.c_void_p(d_r), out.nbytes), "tg_dev_download")
        finally:
            del b_a, b_r
        return (acts, np.ascontiguousarray(out["reward"]), np.ascontiguousarray(out["terminated"]),
                np.ascontiguousarray(out["lines"]))

    def step_sampled(self):
        """Step with the actions left in the device actions buffer by sample_legal_actions()."""
        ptr, _ = self.device_ptr(N.BUF_ACTIONS)
        self.step_device(ptr)

    def features(self):
        """(placement_features u8 [B, 4W, W+3], current_features u8 [B, W+3]) — the observations of
        GroupedActionsObservations(env, observation_wrappers=[FeatureVectorObservation(env)]) and its info["board"]."""
        self._chk(self.lib.tg_features(self.h), "tg_features")
        pf = self._fetch(N.BUF_PLACEMENT_FEATS, np.empty((self.B, 4 * self.W, self.W + 3), np.uint8))
        cf = self._fetch(N.BUF_FEATS, np.empty((self.B, self.W + 3), np.uint8))
        return pf, cf

    def _obs_shapes(self, mode):
        """[(name, per-env shape, dtype)] of observation surface `mode` for this engine's geometry"""
        if mode == N.OBS_DICT:
            return [("board", (self.Hp, self.Wp), np.uint8), ("active_tetromino_mask", (self.Hp, self.Wp), np.uint8),
                    ("holder", (self.P, self.P), np.uint8), ("queue", (self.P, self.P * self.Q), np.uint8)]
        if mode == N.OBS_RGB:
            return [("", (self.Hp, self.Wp + self.P * max(1, self.Q), 3), np.uint8)]
        if mode in (N.OBS_BOARD, N.OBS_FN):
            return [("", (self.H, self.W), np.int8 if mode == N.OBS_FN else np.uint8)]
        raise EngineError(f"no observation surface {mode}")

    def observe_as(self, mode):
        """The current state as observation surface `mode` (N.OBS_BOARD / OBS_FN / OBS_DICT / OBS_RGB), whatever obs_mode the engine
        was created with (tg_observe_as): an array [B, ...] or, for OBS_DICT, the dict of four — what an ObservationWrapper around
        the env returns, without a second engine."""
        fields = self._obs_shapes(mode)
        sizes = [self.B * int(np.prod(sh)) * np.dtype(dt).itemsize for _, sh, dt in fields]
        buf = DeviceBuffer(self, sum(sizes))
        self._chk(self.lib.tg_observe_as(self.h, mode, C.c_void_p(buf.ptr), buf.nbytes), "tg_observe_as")
        raw = self.dev_download(buf.ptr, np.empty(buf.nbytes, np.uint8))
        del buf
        out, o = {}, 0
        for (name, sh, dt), n in zip(fields, sizes):
            out[name] = raw[o:o + n].view(dt).reshape((self.B,) + sh)
            o += n
        return out[""] if "" in out else out

    def features_as(self, placement=False, current=True):
        """(placement_features u8 [B, 4W, W+3] or None, current_features u8 [B, W+3] or None) of the current state for ANY numpy
        rule-set engine (tg_features_as; `features()` is the same for engines created with features=True)."""
        npl, ncu = self.B * 4 * self.W * (self.W + 3), self.B * (self.W + 3)
        buf = DeviceBuffer(self, (npl if placement else 0) + (ncu if current else 0))
        p_pl = C.c_void_p(buf.ptr) if placement else None
        p_cu = C.c_void_p(buf.ptr + (npl if placement else 0)) if current else None
        self._chk(self.lib.tg_features_as(self.h, p_pl, p_cu), "tg_features_as")
        raw = self.dev_download(buf.ptr, np.empty(buf.nbytes, np.uint8))
        del buf
        pl = raw[:npl].reshape(self.B, 4 * self.W, self.W + 3) if placement else None
        cu = raw[(npl if placement else 0):].reshape(self.B, self.W + 3) if current else None
        return pl, cu

    def final_observation(self):
        return self.observation(N.BUF_FINAL_OBS)

    def observe(self):
        """Re-pack the observation buffer from the current state (after set_state)."""
        self._chk(self.lib.tg_observe(self.h), "tg_observe")
        return self.observation()

    # -- state snapshot / restore (envs/tetris.py:681-708 get_state / set_state, batched) --------
    def get_state(self):
        nb, nr, ng = self.B * self.H * self.W, self.B * N.ENV_REC.itemsize, self.B * N.RNG_REC.itemsize
        if nb + nr + ng <= (1 << 16):
            # small handles (the single-env drop-in reads its state after every step): through one page-locked block — three
            # copies into pageable memory are each staged and waited for inside the runtime
            pin = getattr(self, "_pin_state", None)
            if pin is None:
                o1, o2 = (nb + 63) // 64 * 64, (nb + 63) // 64 * 64 + (nr + 63) // 64 * 64
                raw = self.pinned_bytes(o2 + ng)
                pin = self._pin_state = (raw[:nb].reshape(self.B, self.H, self.W), raw[o1:o1 + nr].view(N.ENV_REC), raw[o2:o2 + ng].view(N.RNG_REC))
            self._chk(self.lib.tg_get_state(self.h, N.ptr(pin[0]), N.ptr(pin[1]), N.ptr(pin[2])), "tg_get_state")
            return pin[0].copy(), pin[1].copy(), pin[2].copy()
        board = np.empty((self.B, self.H, self.W), np.uint8)
        rec = np.empty(self.B, N.ENV_REC)
        rng = np.empty(self.B, N.RNG_REC)
        self._chk(self.lib.tg_get_state(self.h, N.ptr(board), N.ptr(rec), N.ptr(rng)), "tg_get_state")
        return board, rec, rng

    def set_state(self, board=None, rec=None, rng=None):
        b = None if board is None else np.ascontiguousarray(board, dtype=np.uint8)
        r = None if rec is None else np.ascontiguousarray(rec, dtype=N.ENV_REC)
        g = None if rng is None else np.ascontiguousarray(rng, dtype=N.RNG_REC)
        if b is not None and b.shape != (self.B, self.H, self.W):
            raise ValueError(f"board must have shape {(self.B, self.H, self.W)}")
        if r is not None and r.shape != (self.B,):
            raise ValueError(f"rec must have shape ({self.B},) of tg_env_rec (got {r.shape})")
        if g is not None and g.shape != (self.B,):
            raise ValueError(f"rng must have shape ({self.B},) of tg_rng_rec (got {g.shape})")
        total = sum(0 if a is None else (a.nbytes + 63) // 64 * 64 for a in (b, r, g))
        if 0 < total <= self._SMALL:  # (small handles: from the page-locked block; tg_set_state waits for its uploads)
            st, o, staged = self._stage(total), 0, []
            for a in (b, r, g):
                if a is None:
                    staged.append(None)
                    continue
                v = st[o:o + a.nbytes]
                v[:] = a.reshape(-1).view(np.uint8)
                staged.append(v.view(a.dtype).reshape(a.shape))
                o += (a.nbytes + 63) // 64 * 64
            b, r, g = staged
        self._chk(self.lib.tg_set_state(self.h, N.ptr(b), N.ptr(r), N.ptr(g)), "tg_set_state")

    def episode_stats(self):
        return self._fetch(N.BUF_EPISODE, np.empty(self.B, N.EPISODE_REC))

    # -- device memory / timing helpers (bench.py) ---------------------------------------------
    def device_ptr(self, which):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.tg_device_ptr(self.h, which, C.byref(p), C.byref(n)), "tg_device_ptr")
        return p.value, n.value

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.tg_dev_alloc(self.h, nbytes, C.byref(p)), "tg_dev_alloc")
        return p.value

    def dev_free(self, p):
        self._chk(self.lib.tg_dev_free(self.h, C.c_void_p(p)), "tg_dev_free")

    def dev_upload(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        self._chk(self.lib.tg_dev_upload(self.h, C.c_void_p(dptr), N.ptr(a), a.nbytes), "tg_dev_upload")

    def dev_download(self, dptr, arr):
        n = arr.nbytes
        if 0 < n <= self._SMALL and arr.flags.c_contiguous:
            st = self._stage(n)
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(st), C.c_void_p(dptr), n), "tg_dev_download")  # (waits for the copy)
            arr.reshape(-1).view(np.uint8)[:] = st
            return arr
        self._chk(self.lib.tg_dev_download(self.h, N.ptr(arr), C.c_void_p(dptr), n), "tg_dev_download")
        return arr

    def sync(self):
        self._chk(self.lib.tg_sync(self.h), "tg_sync")

    def timer_start(self):
        self._chk(self.lib.tg_timer_start(self.h), "tg_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.tg_timer_stop(self.h, C.byref(ms)), "tg_timer_stop")
        return ms.value

    def dict_in_kernel(self) -> bool:
        """True when the step kernel packs this engine's dict / RGB observation itself (FEAT bit 4 / 32 of the instantiation)."""
        if self.obs_mode not in (N.OBS_DICT, N.OBS_RGB):
            return False
        feat = int(self.step_kernel_name().rstrip(">").split(",")[-1])
        return bool(feat & (4 | 32))

    def step_kernel_name(self, transitions_per_launch=1, sampled=False, trajectory=False) -> str:
        """Symbol of the step-kernel instantiation the corresponding entry point launches (as rocprofv3 lists it)."""
        buf = C.create_string_buffer(128)
        self._chk(self.lib.tg_step_kernel_name(self.h, int(transitions_per_launch), int(bool(trajectory)),
                                               int(bool(sampled)), buf, 128), "tg_step_kernel_name")
        return buf.value.decode()

    def stream_copy_gbps(self, nbytes=1 << 30, iters=10) -> float:
        g = C.c_float()
        self._chk(self.lib.tg_stream_copy_gbps(self.h, nbytes, iters, C.byref(g)), "tg_stream_copy_gbps")
        return g.value
