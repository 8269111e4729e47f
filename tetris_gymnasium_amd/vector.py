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

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as N
from ._gym_compat import HAVE_GYMNASIUM, VectorEnvBase, spaces
from .engine import Engine

_OBS = {"dict": N.OBS_DICT, "board": N.OBS_BOARD, "functional": N.OBS_FN, "placement_features": N.OBS_NONE,
        "rgb": N.OBS_RGB, "placement_boards": N.OBS_NONE}
_AUTORESET = {"next_step": N.AUTORESET_NEXT_STEP, "same_step": N.AUTORESET_SAME_STEP, "disabled": N.AUTORESET_NONE}


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


class TetrisVectorEnv(VectorEnvBase):
    metadata = {"render_modes": [], "autoreset_mode": "next_step"}

    def __init__(self, num_envs: int, width: int = 10, height: int = 20, gravity: bool = True, *,
                 ruleset: str = "numpy", observation: str = "dict", queue_size: int | None = None,
                 autoreset_mode: str = "next_step", devices=(0,), episode_stats: bool = True,
                 grouped: bool = False, terminate_on_illegal_action: bool = True, final_observation: bool = False,
                 copy: bool = True):
        """copy=False (gymnasium's SyncVectorEnv(copy=False), gymnasium/vector/sync_vector_env.py): the observations `reset` /
        `step` return are views of rotating page-locked host buffers — valid until the next-but-one call — instead of fresh
        NumPy arrays: the device-to-host copy becomes one DMA at PCIe rate and nothing is allocated per step (single-device
        envs; with several shards the parts are still concatenated into a fresh array).
        final_observation: with autoreset_mode="same_step" and a board / functional observation, `infos` of a step in
        which episodes ended carries "final_observation" (the terminal observation of those envs, the reset observation is
        what `step` returns) and "_final_observation" — gymnasium's same-step autoreset convention
        (examples/train_cnn.py:312-316)."""
        self.num_envs = int(num_envs)
        self.width, self.height = width, height
        self.ruleset = N.RULESET_NUMPY if ruleset == "numpy" else N.RULESET_FUNCTIONAL
        self.obs_kind = observation
        self.autoreset_mode = autoreset_mode
        self.metadata = dict(self.metadata, autoreset_mode=autoreset_mode)
        if queue_size is None:
            queue_size = 4 if self.ruleset == N.RULESET_NUMPY else 7
        self.queue_size = queue_size
        self.ranges = shard_ranges(self.num_envs, len(devices))
        self.engines = [
            Engine(hi - lo, width, height, ruleset=self.ruleset, gravity=gravity, queue_size=queue_size,
                   obs_mode=_OBS[observation], autoreset=_AUTORESET[autoreset_mode], device=dev, env_offset=lo,
                   episode_stats=episode_stats, grouped=grouped, terminate_on_illegal=terminate_on_illegal_action,
                   features=(observation == "placement_features"),
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
            actions = np.asarray(actions, np.int32).reshape(self.num_envs)

        def one(k, e, lo, hi):
            if dev_actions:
                a = actions[k]
                e.step_device(a.ptr if isinstance(a, DeviceArray) else a)
            else:
                e.step_async(actions[lo:hi])
            views = self._dev_views.get(k)  # (the engine's buffers never move: the views are built once per shard)
            if views is None:
                views = self._dev_views[k] = (e.device_array(N.BUF_OBS) if e.obs_mode != N.OBS_NONE else None,
                                              e.device_array(N.BUF_STEP_OUT))
            return views

        jobs = [(k, e, lo, hi) for k, (e, (lo, hi)) in enumerate(zip(self.engines, self.ranges))]
        parts = [one(*j) for j in jobs] if self._pool is None else list(self._pool.map(lambda j: one(*j), jobs))
        return parts[0] if len(parts) == 1 else parts

    def step(self, actions, to_host: bool = True):
        if not to_host:
            return self.step_device(actions)
        actions = np.asarray(actions, np.int32).reshape(self.num_envs)

        def one(e, lo, hi):  # launch, then collect everything this shard returns, on the shard's own thread
            e.step_async(actions[lo:hi])
            out = e.fetch_outputs()
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
            illegal = np.concatenate([o[0] for o in outs]) < 0
            rewards[illegal] = -0.1  # rewards.invalid_action as a Python float, not float32(-0.1)
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

    def close(self, **kwargs):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._dev_views = {}
        for e in self.engines:
            e.close()
        self.engines = []

    close_extras = close
