"""`TetrisVectorEnv`: a gymnasium.vector.VectorEnv adapter over the HIP engine (no reference counterpart —
the reference only ever uses gymnasium's generic SyncVectorEnv: examples/train_ppo.py:185-190).

All `num_envs` environments live on the GPU(s); `step(actions)` is one kernel launch per device.  With
several devices the env batch is sharded by contiguous ranges (env i → device i // (num_envs / n_dev)), no
collective: envs are independent.  Consumes/produces what the reference's trainers use: `reset(seed=...)`,
`step(actions)` → (obs, rewards, terminations, truncations, infos) with `infos["lines_cleared"]`,
`infos["episode"]` / `infos["_episode"]` (RecordEpisodeStatistics convention), `single_action_space`,
`single_observation_space`.
"""
from __future__ import annotations

import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as N
from ._gym_compat import HAVE_GYMNASIUM, VectorEnvBase, spaces
from .engine import Engine

_OBS = {"dict": N.OBS_DICT, "board": N.OBS_BOARD, "functional": N.OBS_FN, "placement_features": N.OBS_NONE,
        "rgb": N.OBS_RGB, "placement_boards": N.OBS_NONE}
_AUTORESET = {"next_step": N.AUTORESET_NEXT_STEP, "same_step": N.AUTORESET_SAME_STEP, "disabled": N.AUTORESET_NONE}


def _gym_autoreset_mode(name: str):
    """metadata["autoreset_mode"] the way the installed gymnasium wants it: gymnasium >= 1.1 (the reference pins 1.1.1,
    poetry.lock:743-744) defines gymnasium.vector.AutoresetMode and its vector wrappers compare / assert against that enum
    (e.g. wrappers.vector.RecordEpisodeStatistics); older versions and the duck-typed fallback take the plain string.
    UNVERIFIED against a live gymnasium — neither image of this project has one (tests/test_gymnasium_conformance.py runs
    wherever `import gymnasium` works)."""
    if HAVE_GYMNASIUM:  # pragma: no cover - depends on the environment
        try:
            from gymnasium.vector import AutoresetMode

            return {"next_step": AutoresetMode.NEXT_STEP, "same_step": AutoresetMode.SAME_STEP,
                    "disabled": AutoresetMode.DISABLED}[name]
        except Exception:
            pass
    return name


def shard_ranges(num_envs: int, n_shards: int):
    """Contiguous, balanced env ranges: shard k owns [lo_k, hi_k).  (env i → GPU i // (B/G))"""
    base, extra = divmod(num_envs, n_shards)
    out, lo = [], 0
    for k in range(n_shards):
        hi = lo + base + (1 if k < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def default_seeds(lo: int, hi: int, seed=None) -> np.ndarray:
    """Env i gets seed `seed + i` (gymnasium vector convention); global index ⇒ results do not depend on
    the number of devices.  seed=None → 1 + i."""
    base = 1 if seed is None else int(seed)
    return np.arange(base + lo, base + hi, dtype=np.uint64)


class _HostStep:
    """One gymnasium-contract step of all shards in ONE call into the C ABI (tg_step_fetch_many): action upload, step launch,
    the device-side pack of rewards / terminations / lines / episode statistics in their final dtypes (TG_BUF_HOST_PACK) and
    the device-to-host copies are enqueued for every shard, then every stream is waited for once.  The observation lands
    where the caller reads it — with one shard the observation and the pack leave as a single copy into one slab laid out
    like the device side; with several shards every array of a shard goes straight into its slice of the whole-batch array —
    and everything `step` returns is a view of that memory: no per-step astype / zeros / concatenate of observation-sized
    arrays.  pinned=True (copy=False): the slab is one of `slots` rotating page-locked buffers (a returned array is valid until
    the next-but-one call).  Otherwise (copy=True) the caller owns what it gets, as with fresh arrays — but a fresh 60 MB array per
    step costs more in page faults than its DMA (65 536 envs, dict observation: 5.5 ms per step against 1.2 ms into page-locked
    memory), so the slabs come from a small pool of page-locked slots and a slot is used again only when NOTHING outside this
    object references any array made from it (`sys.getrefcount` of the slot's root blocks and of every array handed out: a view
    or a tensor made from an array holds a reference to it or to its root).  The usual loop (`obs = next_obs`) alternates between
    two slots; a caller that keeps every observation exhausts the pool (`_POOL_SLOTS`) and gets fresh pageable arrays from then on.
    Replaces the per-env Python loop + np.stack of gymnasium.vector.SyncVectorEnv.step (examples/train_ppo.py:240-248)."""

    def __init__(self, venv, pinned, slots=2, use_pool=True):
        import ctypes as C

        self.C, self.v, self.pinned = C, venv, bool(pinned)
        self.lib = venv.engines[0].lib
        es = venv.engines
        self.n = len(es)
        self.hs = (C.c_void_p * self.n)(*[e.h.value for e in es])
        e0 = es[0]
        # observation arrays: (name, per-env shape, dtype); bytes per env of each
        if e0.obs_mode == N.OBS_DICT:
            self.fields = [("board", (e0.Hp, e0.Wp), np.uint8), ("active_tetromino_mask", (e0.Hp, e0.Wp), np.uint8),
                           ("holder", (e0.P, e0.P), np.uint8), ("queue", (e0.P, e0.P * e0.Q), np.uint8)]
        elif e0.obs_mode == N.OBS_RGB:
            self.fields = [("", (e0.Hp, e0.Wp + e0.P * max(1, e0.Q), 3), np.uint8)]
        else:
            self.fields = [("", (e0.H, e0.W), np.int8 if e0.obs_mode == N.OBS_FN else np.uint8)]
        self.fbytes = [int(np.prod(sh)) * np.dtype(dt).itemsize for _, sh, dt in self.fields]
        # further whole buffers per shard: the legal-action mask of the grouped action set (recomputed on the device for the new
        # state when it is fetched), the terminal observation of same-step autoreset
        self.extras = []
        if venv.grouped:
            self.extras.append(("action_mask", N.BUF_LEGAL_MASK, (4 * e0.W,), np.uint8))
        if venv.final_obs:
            self.extras.append(("final_observation", N.BUF_FINAL_OBS, (e0.H, e0.W), np.int8 if e0.obs_mode == N.OBS_FN else np.uint8))
        self.xbytes = [int(np.prod(sh)) * np.dtype(dt).itemsize for _, _, sh, dt in self.extras]
        self.pack_off = []
        for e in es:
            off = (C.c_size_t * 6)()
            self.lib.tg_host_pack_layout(C.byref(e.cfg), off)
            self.pack_off.append([int(x) for x in off])
        self.single = self.n == 1
        if self.single:  # [TG_BUF_OBS | gap | TG_BUF_HOST_PACK] as on the device: one copy
            po, _ = e0.device_ptr(N.BUF_OBS)
            pp, _ = e0.device_ptr(N.BUF_HOST_PACK)
            self.gap = pp - po
            self.obs_bytes = e0.B * sum(self.fbytes)
            self.m = 2 + len(self.extras)
            self.which = (C.c_int * self.m)(N.BUF_OBS, N.BUF_HOST_PACK, *[x[1] for x in self.extras])
            self.nbytes = (C.c_size_t * self.m)(self.obs_bytes, self.pack_off[0][5], *[e0.B * xb for xb in self.xbytes])
            self.src_off = None
            self.slab_bytes = self.gap + self.pack_off[0][5]
            if not (0 < self.gap - self.obs_bytes <= 16384):  # (not neighbours after all: two copies, still one wait)
                self.gap = (self.obs_bytes + 63) // 64 * 64
                self.slab_bytes = self.gap + self.pack_off[0][5]
        else:  # per shard: every observation array into its slice of the whole-batch array + the shard's pack
            self.m = len(self.fields) + 1 + len(self.extras)
            self.which = (C.c_int * self.m)(*([N.BUF_OBS] * len(self.fields) + [N.BUF_HOST_PACK] + [x[1] for x in self.extras]))
            nb, so = [], []
            for e, offs in zip(es, self.pack_off):
                o = 0
                for fb in self.fbytes:
                    nb.append(e.B * fb); so.append(o); o += e.B * fb
                nb.append(offs[5]); so.append(0)
                for xb in self.xbytes:
                    nb.append(e.B * xb); so.append(0)
            self.nbytes = (C.c_size_t * (self.n * self.m))(*nb)
            self.src_off = (C.c_size_t * (self.n * self.m))(*so)
        self.slots, self.cur = [], 0
        self.n_slots = max(2, int(slots)) if self.pinned else 0
        # copy=True: the refcount-guarded pool (CPython with the GIL only — a free-threaded build's reference counts are not a usable
        # ownership test; at most _POOL_BYTES of page-locked memory).  copy="fresh" switches it off: every call returns newly allocated
        # pageable arrays, as gymnasium's SyncVectorEnv(copy=True) does.
        gil = getattr(sys, "_is_gil_enabled", lambda: True)()
        self.pool = [] if (not self.pinned and use_pool and gil and hasattr(sys, "getrefcount")) else None
        self.pool_slots = 0
        if self.pool is not None:
            per_slot = max(1, sum(int(b) for b in self.nbytes))
            self.pool_slots = min(self._POOL_SLOTS, self._POOL_BYTES // per_slot)
        self.act_ptrs = (C.c_void_p * self.n)()
        self._act_stage = None

    # -- destination memory ---------------------------------------------------------------------------------------------
    _POOL_SLOTS = 3
    _POOL_BYTES = 1 << 30

    def _raw(self, nbytes, pinned):
        """uint8[nbytes]: page-locked or a fresh pageable array"""
        if not pinned:
            return np.empty(nbytes, np.uint8)
        return self.v.engines[0].pinned_bytes(nbytes)  # (frees itself when the last view dies: valid beyond close())

    @staticmethod
    def _refs(watch):
        return [sys.getrefcount(a) for a in watch]

    def _make_slot(self, pinned):
        C, v = self.C, self.v
        B = v.num_envs
        slot = {}
        raws = []  # every block of destination memory of this slot

        def _raw(nbytes):
            raws.append(self._raw(nbytes, pinned))
            return raws[-1]

        dst = (C.c_void_p * (self.n * self.m))()
        if self.single:
            raw = _raw(self.slab_bytes)
            base = raw.ctypes.data
            dst[0], dst[1] = base, base + self.gap
            obs, o = {}, 0
            for (name, sh, dt), fb in zip(self.fields, self.fbytes):
                obs[name] = raw[o:o + B * fb].view(dt).reshape((B,) + sh)
                o += B * fb
            packs = [raw[self.gap:self.gap + self.pack_off[0][5]]]
            xs = {}
            for i, ((name, _, sh, dt), xb) in enumerate(zip(self.extras, self.xbytes)):
                xr = _raw(B * xb)
                dst[2 + i] = xr.ctypes.data
                xs[name] = xr.view(dt).reshape((B,) + sh)
            slot["keep"] = raw
        else:
            obs, keep = {}, []
            for (name, sh, dt), fb in zip(self.fields, self.fbytes):
                raw = _raw(B * fb)
                keep.append(raw)
                obs[name] = raw.view(dt).reshape((B,) + sh)
            packs = []
            for k, ((lo, hi), offs) in enumerate(zip(v.ranges, self.pack_off)):
                for f, fb in enumerate(self.fbytes):
                    dst[k * self.m + f] = keep[f].ctypes.data + lo * fb
                pk = _raw(offs[5])
                packs.append(pk)
                dst[k * self.m + len(self.fbytes)] = pk.ctypes.data
            xs = {}
            for i, ((name, _, sh, dt), xb) in enumerate(zip(self.extras, self.xbytes)):
                xr = _raw(B * xb)
                keep.append(xr)
                xs[name] = xr.view(dt).reshape((B,) + sh)
                for k, (lo, hi) in enumerate(v.ranges):
                    dst[k * self.m + len(self.fbytes) + 1 + i] = xr.ctypes.data + lo * xb
            slot["keep"] = keep
        slot["dst"] = dst
        slot["extras"] = xs
        slot["obs"] = obs[""] if "" in obs else obs
        # the five arrays of every shard's pack, as views
        parts = []
        for pk, offs, (lo, hi) in zip(packs, self.pack_off, v.ranges):
            b = hi - lo
            # (rewards, terminations, lines, episode returns, episode lengths) — tg_host_pack_layout's offsets 0, 4, 2, 1, 3
            parts.append((pk[offs[0]:offs[0] + 8 * b].view(np.float64), pk[offs[4]:offs[4] + b].view(np.bool_),
                          pk[offs[2]:offs[2] + 4 * b].view(np.int32), pk[offs[1]:offs[1] + 8 * b].view(np.float64),
                          pk[offs[3]:offs[3] + 4 * b].view(np.int32)))
        slot["parts"] = parts
        if pinned and self.pool is not None:
            # what must be unreferenced from outside before the slot is written again: the root arrays of its blocks (a view of a
            # view keeps the ROOT as its base) and every array this object hands out
            watch = [r.base for r in raws if isinstance(r.base, np.ndarray)] + raws
            watch += list(obs.values()) + list(xs.values()) + [a for part in parts for a in part]
            slot["watch"] = watch
        return slot

    def _next_slot(self):
        if not self.pinned:
            if self.pool is None:
                return self._make_slot(False)
            for slot in self.pool:
                if self._refs(slot["watch"]) == slot["idle"]:
                    return slot
            if len(self.pool) < self.pool_slots:
                slot = self._make_slot(True)
                slot["idle"] = self._refs(slot["watch"])  # (the reference counts with nothing handed out, taken where they are compared)
                self.pool.append(slot)
                return slot
            return self._make_slot(False)  # everything handed out so far is still held: fresh pageable arrays
        if len(self.slots) < self.n_slots:
            self.slots.append(self._make_slot(True))
            self.cur = len(self.slots) - 1
            return self.slots[-1]
        self.cur = (self.cur + 1) % self.n_slots
        return self.slots[self.cur]

    # -- the call ---------------------------------------------------------------------------------------------------------
    def step(self, actions):
        """actions: contiguous int32[num_envs] (host) or None (fetch only).  Returns (obs, rewards f64, terminations bool,
        lines i32, episode returns f64, episode lengths i32, extras {"action_mask", "final_observation"}) — whole-batch arrays."""
        v, C = self.v, self.C
        slot = self._next_slot()
        acts = None
        if actions is not None:
            # through a page-locked copy: an upload from pageable memory is staged and waited for inside the runtime (the call waits
            # for every stream before it returns, so the block is free again at the next step)
            if self._act_stage is None:
                self._act_stage = v.engines[0].pinned_bytes(4 * v.num_envs).view(np.int32)
            self._act_stage[:] = actions
            base = self._act_stage.ctypes.data
            for k, (lo, _) in enumerate(v.ranges):
                self.act_ptrs[k] = base + 4 * lo
            acts = self.act_ptrs
        rc = self.lib.tg_step_fetch_many(self.n, self.hs, acts, N.MEM_HOST, self.m, self.which, slot["dst"], self.nbytes,
                                         self.src_off)
        if rc != 0:
            for e in v.engines:  # (the failing handle carries the message)
                msg = self.lib.tg_last_error(e.h).decode()
                if msg:
                    raise RuntimeError(f"tg_step_fetch_many failed ({rc}): {msg}")
            raise RuntimeError(f"tg_step_fetch_many failed ({rc})")
        parts = slot["parts"]
        obs = slot["obs"]
        obs = dict(obs) if isinstance(obs, dict) else obs  # (the caller's own dict)
        if len(parts) == 1:
            return (obs,) + parts[0] + (dict(slot["extras"]),)
        return (obs,) + tuple(np.concatenate([p[i] for p in parts]) for i in range(5)) + (dict(slot["extras"]),)  # (25 bytes per env)


class TetrisVectorEnv(VectorEnvBase):
    metadata = {"render_modes": [], "autoreset_mode": "next_step"}

    def __init__(self, num_envs: int, width: int = 10, height: int = 20, gravity: bool = True, *,
                 ruleset: str = "numpy", observation: str = "dict", queue_size: int | None = None,
                 autoreset_mode: str = "next_step", devices=(0,), episode_stats: bool = True,
                 grouped: bool = False, terminate_on_illegal_action: bool = True, final_observation: bool = False,
                 copy: "bool | str" = True, actions_mapping=None, rewards_mapping=None):
        """copy=False (gymnasium's SyncVectorEnv(copy=False), gymnasium/vector/sync_vector_env.py): the observations `reset` /
        `step` return are views of rotating page-locked host buffers — valid until the next-but-one call — instead of fresh
        NumPy arrays: the device-to-host copy becomes one DMA at PCIe rate and nothing is allocated per step (single-device
        envs; with several shards the parts are still concatenated into a fresh array).
        copy=True (default): the caller owns what it gets.  The arrays come from a small pool of page-locked slots (at most 1 GiB;
        a slot is written again only when no Python object references any array made from it: `sys.getrefcount`).  OWNERSHIP CAVEAT:
        that test sees Python references only.  Memory that is still being read WITHOUT one — `torch.from_numpy(obs).to(dev,
        non_blocking=True)` from this page-locked memory is a genuinely asynchronous DMA; a raw pointer taken through ctypes — must
        be kept alive by its reader (hold `obs` until the copy has completed: `torch.cuda.current_stream().synchronize()` or an
        event), or use copy="fresh": newly allocated pageable arrays every call, exactly gymnasium's copy=True, at the price of the
        page faults (5.5 vs 1.2 ms per step for 65 536 dict observations).  The pool is off on free-threaded Python builds.
        final_observation: with autoreset_mode="same_step" and a board / functional observation, `infos` of a step in
        which episodes ended carries "final_observation" (the terminal observation of those envs, the reset observation is
        what `step` returns) and "_final_observation" — gymnasium's same-step autoreset convention
        (examples/train_cnn.py:312-316).
        actions_mapping / rewards_mapping (numpy rule-set): what `Tetris(actions_mapping=..., rewards_mapping=...)` takes
        (reference envs/tetris.py:83-84); the rewards `step` returns are the Python values (int score + the mapping's floats)."""
        self.num_envs = int(num_envs)
        self.width, self.height = width, height
        self.ruleset = N.RULESET_NUMPY if ruleset == "numpy" else N.RULESET_FUNCTIONAL
        self.obs_kind = observation
        self.autoreset_mode = autoreset_mode
        self.metadata = dict(self.metadata, autoreset_mode=_gym_autoreset_mode(autoreset_mode))
        if queue_size is None:
            queue_size = 4 if self.ruleset == N.RULESET_NUMPY else 7
        self.queue_size = queue_size
        self.ranges = shard_ranges(self.num_envs, len(devices))
        self.engines = [
            Engine(hi - lo, width, height, ruleset=self.ruleset, gravity=gravity, queue_size=queue_size,
                   obs_mode=_OBS[observation], autoreset=_AUTORESET[autoreset_mode], device=dev, env_offset=lo,
                   episode_stats=episode_stats, grouped=grouped, terminate_on_illegal=terminate_on_illegal_action,
                   features=(observation == "placement_features"),
                   actions_mapping=actions_mapping, rewards_mapping=rewards_mapping,
                   final_obs=(bool(final_observation) and autoreset_mode == "same_step" and observation in ("board", "functional")))
            for (lo, hi), dev in zip(self.ranges, devices) if hi > lo
        ]
        self.ranges = [r for r in self.ranges if r[1] > r[0]]
        self.episode_stats = episode_stats
        self.grouped = grouped  # wrappers/grouped.py action set: column * 4 + rotations, infos["action_mask"]
        n_act = 4 * width if grouped else (8 if self.ruleset == N.RULESET_NUMPY else 7)
        hp, wp, p = height + 4, width + 8, 4
        self.single_action_space = spaces.Discrete(n_act)
        if observation == "dict":
            self.single_observation_space = spaces.Dict({
                "board": spaces.Box(low=0, high=9, shape=(hp, wp), dtype=np.uint8),
                "active_tetromino_mask": spaces.Box(low=0, high=1, shape=(hp, wp), dtype=np.uint8),
                "holder": spaces.Box(low=0, high=9, shape=(p, p), dtype=np.uint8),
                "queue": spaces.Box(low=0, high=9, shape=(p, p * queue_size), dtype=np.uint8),
            })
        elif observation == "board":
            self.single_observation_space = spaces.Box(low=0, high=9, shape=(height, width), dtype=np.uint8)
        elif observation == "placement_features":
            # GroupedActionsObservations(env, observation_wrappers=[FeatureVectorObservation(env)]) (train_lin_grouped.py)
            if not grouped:
                raise ValueError("observation='placement_features' needs grouped=True")
            self.single_observation_space = spaces.Box(low=0, high=255, shape=(4 * width, width + 3), dtype=np.uint8)
        elif observation == "rgb":
            # RgbObservation(env) (wrappers/observation.py:11-74; examples/train_ppo.py:96-110)
            self.single_observation_space = spaces.Box(low=0, high=255, shape=(hp, wp + p * max(1, queue_size), 3),
                                                       dtype=np.uint8)
        elif observation == "placement_boards":
            # GroupedActionsObservations(env) without observation wrappers (wrappers/grouped.py:76-84, 124-207)
            if not grouped:
                raise ValueError("observation='placement_boards' needs grouped=True")
            self.single_observation_space = spaces.Box(low=0, high=9, shape=(4 * width, hp, wp), dtype=np.uint8)
        else:
            self.single_observation_space = spaces.Box(low=-1, high=1, shape=(height, width), dtype=np.int8)
        if HAVE_GYMNASIUM:  # pragma: no cover - depends on the environment
            from gymnasium.vector.utils import batch_space

            self.action_space = batch_space(self.single_action_space, self.num_envs)
            self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        else:
            self.action_space = spaces.MultiDiscrete([n_act] * self.num_envs)
            self.observation_space = None
        self._prev_len = np.zeros(self.num_envs, np.uint32)
        # one worker per engine (= per device): launches, fetches and the NumPy unpacking of a shard run beside the other
        # shards' (ctypes releases the GIL inside the C ABI).  _each() submits exactly one job per engine and waits for all
        # of them, so calls on ONE handle never overlap (the C ABI's contract); which pool thread runs a shard does not
        # matter — every entry point sets its device itself.
        self._pool = ThreadPoolExecutor(len(self.engines), thread_name_prefix="tg-shard") if len(self.engines) > 1 else None
        self._dev_views = {}  # shard -> (observation view(s), outputs views) of step_device
        if isinstance(copy, str) and copy != "fresh":
            raise ValueError("copy: True, False or 'fresh'")
        self._copy, self._host, self._sd = bool(copy), None, None
        self._fresh = copy == "fresh"
        self._ones, self._falses = np.ones(self.num_envs, bool), np.zeros(self.num_envs, bool)
        self._zeros_t = np.zeros(self.num_envs, np.float64)  # infos["episode"]["t"] (elapsed time: not tracked)
        if not copy:
            for e in self.engines:
                e.use_pinned_observations(2)
        self.final_obs = bool(final_observation) and autoreset_mode == "same_step" and observation in ("board", "functional")

    # ---- helpers --------------------------------------------------------------------------------
    def _each(self, fn):
        """fn(engine, lo, hi) for every shard, each on its own thread; results in shard order."""
        jobs = [(e, lo, hi) for e, (lo, hi) in zip(self.engines, self.ranges)]
        if self._pool is None:
            return [fn(*j) for j in jobs]
        return list(self._pool.map(lambda j: fn(*j), jobs))

    def _shard_obs(self, e):
        """(observation of one shard, its current feature vectors or None)"""
        if self.obs_kind == "placement_features":
            pf, cf = e.features()
            return pf, cf
        if self.obs_kind == "placement_boards":
            return e.placement_boards()[1], None
        return e.observation(), None

    def _join_obs(self, parts):
        if self.obs_kind == "placement_features":
            self._cur_feats = np.concatenate([p[1] for p in parts])
        obs = [p[0] for p in parts]
        if len(obs) == 1:  # one shard: its arrays ARE the result (a concatenate of one part would copy 13 MB per step for nothing)
            return obs[0]
        if isinstance(obs[0], dict):
            return {k: np.concatenate([o[k] for o in obs]) for k in obs[0]}
        return np.concatenate(obs)

    # ---- gymnasium.vector API ---------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        mask = None if not options else options.get("reset_mask")
        self._reset_count = getattr(self, "_reset_count", 0) + 1

        def one(e, lo, hi):
            if isinstance(seed, (list, tuple, np.ndarray)):
                seeds = np.asarray(seed, np.uint64)[lo:hi]
            elif seed is None and self.ruleset == N.RULESET_NUMPY:
                seeds = None  # keep running generators (never-seeded envs get 1 + global index)
            elif seed is None:
                # functional rule-set: the key of a reset is an explicit input (envs/tetris_fn.py:318-324); an unseeded
                # reset must not replay the previous episodes, so every unseeded reset derives fresh per-env keys
                seeds = default_seeds(lo, hi, 1 + (self._reset_count - 1) * self.num_envs)
            else:
                seeds = default_seeds(lo, hi, seed)
            e.reset(seeds, None if mask is None else np.asarray(mask, np.uint8)[lo:hi])
            return (e.legal_mask() if self.grouped else None), self._shard_obs(e)

        parts = self._each(one)
        self._prev_len[:] = 0
        info = {"lines_cleared": np.zeros(self.num_envs, np.int32)}
        if self.grouped:
            info["action_mask"] = np.concatenate([p[0] for p in parts])
        obs = self._join_obs([p[1] for p in parts])
        if self.obs_kind == "placement_features":
            info["board"] = self._cur_feats
        return obs, info

    def step_device(self, actions):
        """One transition whose results STAY ON THE DEVICE: returns (observations, outputs) as zero-copy DeviceArray views of
        the engines' buffers (`__cuda_array_interface__` / DLPack: `torch.as_tensor(v, device="cuda")`) — observations in the
        shapes `step` returns (a dict of views for the dict observation), outputs = {"reward" f32, "terminated" u8, "lines"
        u8}; with several devices one (observations, outputs) pair per shard, in env order.  `actions`: a host array, or
        per shard a DeviceArray / device pointer to int32[shard size].  Nothing is copied to the host and nothing waits for
        the GPU; the views alias buffers the next transition overwrites.  What a trainer's rollout loop needs instead of
        `torch.Tensor(next_obs).to(device)` (examples/train_ppo.py:240-245)."""
        from .engine import DeviceArray

        if isinstance(actions, (DeviceArray, int)):
            actions = [actions]
        dev_actions = isinstance(actions, (list, tuple)) and len(actions) == len(self.engines) and \
            all(isinstance(a, (DeviceArray, int)) for a in actions)
        if not dev_actions:
            actions = np.ascontiguousarray(actions, np.int32).reshape(self.num_envs)

        # every shard's launch from ONE call into the C ABI (tg_step_fetch_many with nothing to fetch): one handle after the
        # other on this thread — a launch is an asynchronous ~10 us enqueue, a hand-off to a worker thread per shard under the
        # GIL costs more than that (round 3 used the thread pool here: 6.8 us per step with one shard, but growing with k)
        import ctypes as C

        if self._sd is None:
            self._sd = ((C.c_void_p * len(self.engines))(*[e.h.value for e in self.engines]), (C.c_void_p * len(self.engines))())
        hs, ptrs = self._sd
        if dev_actions:
            for k, a in enumerate(actions):
                ptrs[k] = a.ptr if isinstance(a, DeviceArray) else a
        else:
            base = actions.ctypes.data
            for k, (lo, _) in enumerate(self.ranges):
                ptrs[k] = base + 4 * lo
        lib = self.engines[0].lib
        rc = lib.tg_step_fetch_many(len(self.engines), hs, ptrs, N.MEM_DEVICE if dev_actions else N.MEM_HOST, 0, None, None, None, None)
        if rc != 0:
            msgs = [lib.tg_last_error(e.h).decode() for e in self.engines]
            raise RuntimeError(f"tg_step_fetch_many failed ({rc}): {next((m for m in msgs if m), '')}")
        parts = []
        for k, e in enumerate(self.engines):
            views = self._dev_views.get(k)  # (the engine's buffers never move: the views are built once per shard)
            if views is None:
                views = self._dev_views[k] = (e.device_array(N.BUF_OBS) if e.obs_mode != N.OBS_NONE else None,
                                              e.device_array(N.BUF_STEP_OUT))
            parts.append(views)
        return parts[0] if len(parts) == 1 else parts

    def step(self, actions, to_host: bool = True):
        if not to_host:
            return self.step_device(actions)
        actions = np.ascontiguousarray(actions, np.int32).reshape(self.num_envs)
        if self.obs_kind in ("board", "functional", "dict", "rgb"):
            # the whole step — upload, launch, pack, copies, one wait per device — is one call into the C ABI
            if self._host is None:
                self._host = _HostStep(self, pinned=not self._copy, use_pool=not self._fresh)
            obs, rewards, terminations, lines, ep_r, ep_l, extras = self._host.step(actions)
            fresh = (lambda a: a.copy()) if self._copy else (lambda a: a)  # (copy=False: the constant arrays are shared, too)
            infos = {"lines_cleared": lines, "_lines_cleared": fresh(self._ones)}
            any_done = terminations.any()
            if self.episode_stats and any_done:
                infos["episode"] = {"r": ep_r, "l": ep_l, "t": fresh(self._zeros_t)}
                infos["_episode"] = fresh(terminations)
            if self.grouped:  # the mask the NEXT action must be drawn from (wrappers/grouped.py:266-267)
                infos["action_mask"] = extras["action_mask"]
            if self.final_obs and any_done:  # (valid where `_final_observation`; the other rows are whatever they last held)
                infos["final_observation"] = extras["final_observation"]
                infos["_final_observation"] = fresh(terminations)
            return obs, rewards, terminations, fresh(self._falses), infos

        def one(e, lo, hi):  # launch, then collect everything this shard returns, on the shard's own thread
            e.step_async(actions[lo:hi])
            so = e.fetch_step_out()
            out = (np.ascontiguousarray(so["reward"]), np.ascontiguousarray(so["terminated"]), np.ascontiguousarray(so["lines"]),
                   np.ascontiguousarray(so["flags"]))
            ep = e.episode_stats() if (self.episode_stats and out[1].any()) else None
            lm = e.legal_mask() if self.grouped else None
            fo = e.final_observation() if (self.final_obs and out[1].any()) else None
            return out, ep, lm, self._shard_obs(e), fo

        parts = self._each(one)
        outs = [p[0] for p in parts]
        rewards = np.concatenate([o[0] for o in outs]).astype(np.float64)
        terminations = np.concatenate([o[1] for o in outs]).astype(bool)
        lines = np.concatenate([o[2] for o in outs]).astype(np.int32)
        infos = {"lines_cleared": lines, "_lines_cleared": np.ones(self.num_envs, bool)}
        if self.episode_stats and terminations.any():
            r = np.zeros(self.num_envs, np.float64)
            ln = np.zeros(self.num_envs, np.int32)
            for (lo, hi), p in zip(self.ranges, parts):
                if p[1] is not None:
                    t = terminations[lo:hi]
                    r[lo:hi] = np.where(t, p[1]["last_ret"], 0.0)
                    ln[lo:hi] = np.where(t, p[1]["last_len"], 0)
            infos["episode"] = {"r": r, "l": ln}
            infos["_episode"] = terminations.copy()
        if self.grouped:  # the mask the NEXT action must be drawn from (wrappers/grouped.py:266-267)
            infos["action_mask"] = np.concatenate([p[2] for p in parts])
            illegal = (np.concatenate([o[3] for o in outs]) & N.OUT_ILLEGAL) != 0
            rewards[illegal] = self.engines[0].rewards[3]  # rewards.invalid_action as a Python float, not float32(-0.1)
        if self.engines[0].cfg.flags & N.FLAG_CUSTOM_REWARDS:  # what Python computes: int score + the mapping's floats (envs/tetris.py:461-474)
            fl = np.concatenate([o[3] for o in outs])
            al, _, go, inv = self.engines[0].rewards
            com = (fl & N.OUT_COMMITTED) != 0
            rewards = np.where((fl & N.OUT_ILLEGAL) != 0, inv,
                               np.where(com, np.where(terminations, go, (lines.astype(np.int64) ** 2 * self.width).astype(np.float64) + al), 0.0))
        truncations = np.zeros(self.num_envs, bool)
        if self.final_obs and terminations.any():  # (valid where terminated; zeros elsewhere)
            fo = np.zeros((self.num_envs, self.height, self.width), np.int8 if self.obs_kind == "functional" else np.uint8)
            for (lo, hi), p in zip(self.ranges, parts):
                if p[4] is not None:
                    fo[lo:hi][terminations[lo:hi]] = p[4][terminations[lo:hi]]
            infos["final_observation"] = fo
            infos["_final_observation"] = terminations.copy()
        obs = self._join_obs([p[3] for p in parts])
        if self.obs_kind == "placement_features":
            infos["board"] = self._cur_feats  # wrappers/grouped.py:261-264 info["board"]
        return obs, rewards, terminations, truncations, infos

    def rollout(self, actions):
        """K transitions for pre-generated actions [K][num_envs] in one call per device (tg_rollout: up to 16 transitions
        per launch, state resident on chip).  Returns (obs[K][num_envs]..., rewards, terminations, lines) — the
        trajectory a `for k: envs.step(actions[k])` loop would have produced (autoreset as configured)."""
        actions = np.asarray(actions)
        n_act = self.single_action_space.n
        if actions.ndim != 2 or actions.shape[1] != self.num_envs:
            raise ValueError("actions must have shape (K, num_envs)")
        if self.ruleset == N.RULESET_NUMPY and ((actions < 0) | (actions >= n_act)).any():
            raise ValueError(f"action outside Discrete({n_act})")  # envs/tetris.py:215-217 asserts action_space.contains
        actions = actions.astype(np.int8)
        parts = self._each(lambda e, lo, hi: e.rollout(actions[:, lo:hi]))
        obs = None if parts[0][0] is None else np.concatenate([p[0] for p in parts], axis=1)
        return (obs, np.concatenate([p[1] for p in parts], axis=1).astype(np.float64),
                np.concatenate([p[2] for p in parts], axis=1).astype(bool),
                np.concatenate([p[3] for p in parts], axis=1).astype(np.int32))

    def rollout_device(self, actions, K: int, obs_traj=None, out_traj=None, chunk: int = 0):
        """K transitions per shard between DEVICE buffers, enqueued and not waited for: per shard `actions[k]` = device pointer /
        DeviceArray of int8[K][shard size], `obs_traj[k]` / `out_traj[k]` (optional) where the observation / tg_step_out record of
        every transition go (slot j at j * shard size * bytes per env).  `chunk` > 0 enqueues the K transitions in pieces of that
        many (a multiple of the engine's transitions per launch keeps every launch full) round-robin over the shards, so that every
        device has work queued while the host walks the others.  The regime for ONE process driving many devices (VERDICT r05 weak
        #8): a one-transition `step(..., to_host=False)` costs the host ~1.5 us per shard and launch, i.e. ~12 us per step for eight
        shards against kernels of ~6 us — driven this way the host pays one enqueue per shard and 16 transitions."""
        import ctypes as C

        from .engine import DeviceArray

        ptr = lambda a: None if a is None else (a.ptr if isinstance(a, DeviceArray) else int(a))  # noqa: E731
        n = len(self.engines)
        if len(actions) != n or (obs_traj is not None and len(obs_traj) != n) or (out_traj is not None and len(out_traj) != n):
            raise ValueError(f"one device buffer per shard ({n})")
        K = int(K)
        step = int(chunk) if chunk and chunk > 0 else K
        for k0 in range(0, K, step):
            kk = min(step, K - k0)
            for i, e in enumerate(self.engines):
                b = e.B
                per_env = e.lib.tg_obs_bytes_per_env(C.byref(e.cfg)) if (obs_traj is not None and obs_traj[i] is not None) else 0
                a = ptr(actions[i]) + k0 * b
                o = None if not per_env else ptr(obs_traj[i]) + k0 * b * per_env
                r = None if (out_traj is None or out_traj[i] is None) else ptr(out_traj[i]) + k0 * b * N.STEP_OUT.itemsize
                e.rollout_device(a, kk, o, r)

    # ---- gymnasium.vector.VectorEnv odds and ends (call / get_attr / set_attr of the sub-environments) ----------------
    def get_attr(self, name: str):
        """Per-env values of an attribute, as gymnasium.vector.VectorEnv.get_attr returns them (a tuple of num_envs entries).
        The sub-environments are rows of the device batch, not Python objects: the attributes that exist are the
        configuration (`width`, `height`, `queue_size`, `gravity_enabled`, `padding`, `render_mode`, `metadata`, `spec`)."""
        const = {"width": self.width, "height": self.height, "queue_size": self.queue_size, "padding": 4,
                 "gravity_enabled": bool(self.engines[0].cfg.gravity) if self.engines else True, "render_mode": None,
                 "metadata": {"render_modes": []}, "spec": None}
        if name not in const:
            raise AttributeError(f"the sub-environments of TetrisVectorEnv have no attribute {name!r} (they live on the GPU)")
        return tuple(const[name] for _ in range(self.num_envs))

    def set_attr(self, name: str, values):
        raise AttributeError(f"cannot set {name!r}: the sub-environments of TetrisVectorEnv are rows of a device batch; "
                             "create a new TetrisVectorEnv with the configuration you want")

    def call(self, name: str, *args, **kwargs):
        """gymnasium.vector.VectorEnv.call: attributes come from get_attr; methods of a sub-environment do not exist here."""
        return self.get_attr(name)

    def close(self, **kwargs):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._dev_views = {}
        self._host = None
        for e in self.engines:
            e.close()
        self.engines = []

    close_extras = close
