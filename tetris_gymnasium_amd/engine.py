"""`Engine`: the batched device-resident Tetris state behind the C ABI (one handle = one GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


class EngineError(RuntimeError):
    pass


class Engine:
    """Owns `batch` environments on one MI355X.  All arrays in/out are NumPy; state stays on the device.

    ruleset NUMPY mirrors tetris_gymnasium/envs/tetris.py (Tetris.step / reset), FUNCTIONAL mirrors
    tetris_gymnasium/envs/tetris_fn.py (step / reset).
    """

    def __init__(self, batch, width=10, height=20, *, ruleset=N.RULESET_NUMPY, gravity=True, queue_size=None,
                 obs_mode=None, autoreset=N.AUTORESET_NONE, device=0, env_offset=0, padding=4, episode_stats=False,
                 final_obs=False, grouped=False, terminate_on_illegal=True, features=False):
        self.lib = N.load()
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

    def rollout(self, actions, want_obs=True):
        """K transitions with the given actions int8[K][B] in one call (tg_rollout: the default kernel runs up to 8 per
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
        d_a = self.dev_alloc(a.nbytes)
        d_o = self.dev_alloc(K * self.B * per_env) if (per_env and want_obs) else None
        d_r = self.dev_alloc(K * self.B * N.STEP_OUT.itemsize)
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
            self.dev_free(d_a)
            if d_o:
                self.dev_free(d_o)
            self.dev_free(d_r)
        return (obs, np.ascontiguousarray(out["reward"]), np.ascontiguousarray(out["terminated"]),
                np.ascontiguousarray(out["lines"]))

    def step(self, actions):
        """One transition; returns (reward f32[B], terminated u8[B], lines u8[B])."""
        self.step_async(actions)
        return self.fetch_outputs()

    def fetch_outputs(self):
        out = np.empty(self.B, N.STEP_OUT)
        self._chk(self.lib.tg_fetch(self.h, N.BUF_STEP_OUT, N.ptr(out), out.nbytes), "tg_fetch(step_out)")
        return (np.ascontiguousarray(out["reward"]), np.ascontiguousarray(out["terminated"]),
                np.ascontiguousarray(out["lines"]))

    # -- observations --------------------------------------------------------------------------
    def _fetch(self, which, arr):
        self._chk(self.lib.tg_fetch(self.h, which, N.ptr(arr), arr.nbytes), f"tg_fetch({which})")
        return arr

    def observation(self, which=N.BUF_OBS):
        """The observation buffer of the engine's obs_mode as NumPy array(s)."""
        if self.obs_mode == N.OBS_BOARD:
            return self._fetch(which, np.empty((self.B, self.H, self.W), np.uint8))
        if self.obs_mode == N.OBS_FN:
            return self._fetch(which, np.empty((self.B, self.H, self.W), np.int8))
        if self.obs_mode == N.OBS_DICT:
            cells, pp, B = self.Hp * self.Wp, self.P * self.P, self.B
            flat = self._fetch(which, np.empty(B * (2 * cells + pp + pp * self.Q), np.uint8))
            o = 0
            board = flat[o:o + B * cells].reshape(B, self.Hp, self.Wp); o += B * cells
            mask = flat[o:o + B * cells].reshape(B, self.Hp, self.Wp); o += B * cells
            holder = flat[o:o + B * pp].reshape(B, self.P, self.P); o += B * pp
            queue = flat[o:o + B * pp * self.Q].reshape(B, self.P, self.P * self.Q)
            return {"board": board, "active_tetromino_mask": mask, "holder": holder, "queue": queue}
        if self.obs_mode == N.OBS_RGB:
            return self._fetch(which, np.empty((self.B, self.Hp, self.Wp + self.P * max(1, self.Q), 3), np.uint8))
        raise EngineError("engine was created with obs_mode NONE")

    # -- grouped action surface (wrappers/grouped.py) ---------------------------------------------
    def legal_mask(self):
        """u8 [B, 4*W]: 1 = legal placement (column*4 + rotations) for the current active pieces."""
        self._chk(self.lib.tg_grouped_enumerate(self.h, None), "tg_grouped_enumerate")
        return self._fetch(N.BUF_LEGAL_MASK, np.empty((self.B, 4 * self.W), np.uint8))

    def placement_boards(self):
        """(legal mask u8[B][4W], boards u8[B][4W][H+p][W+2p]): the observation of GroupedActionsObservations
        (wrappers/grouped.py:124-207) for the current state, built on the device."""
        n = self.B * 4 * self.W * self.Hp * self.Wp
        d = self.dev_alloc(n)
        try:
            self._chk(self.lib.tg_grouped_enumerate(self.h, C.c_void_p(d)), "tg_grouped_enumerate(boards)")
            boards = np.empty((self.B, 4 * self.W, self.Hp, self.Wp), np.uint8)
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(boards), C.c_void_p(d), n), "tg_dev_download")
        finally:
            self.dev_free(d)
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
        d_a = self.dev_alloc(K * self.B)
        d_r = self.dev_alloc(K * self.B * N.STEP_OUT.itemsize)
        try:
            self._chk(self.lib.tg_grouped_rollout(self.h, int(seed) & 0xFFFFFFFF, int(counter0) & 0xFFFFFFFF, int(K),
                                                  C.c_void_p(d_a), None, C.c_void_p(d_r)), "tg_grouped_rollout")
            acts = np.empty((K, self.B), np.int8)
            out = np.empty((K, self.B), N.STEP_OUT)
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(acts), C.c_void_p(d_a), acts.nbytes), "tg_dev_download")
            self._chk(self.lib.tg_dev_download(self.h, N.ptr(out), C.c_void_p(d_r), out.nbytes), "tg_dev_download")
        finally:
            self.dev_free(d_a)
            self.dev_free(d_r)
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

    def final_observation(self):
        return self.observation(N.BUF_FINAL_OBS)

    def observe(self):
        """Re-pack the observation buffer from the current state (after set_state)."""
        self._chk(self.lib.tg_observe(self.h), "tg_observe")
        return self.observation()

    # -- state snapshot / restore (envs/tetris.py:681-708 get_state / set_state, batched) --------
    def get_state(self):
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
        self._chk(self.lib.tg_dev_download(self.h, N.ptr(arr), C.c_void_p(dptr), arr.nbytes), "tg_dev_download")
        return arr

    def sync(self):
        self._chk(self.lib.tg_sync(self.h), "tg_sync")

    def timer_start(self):
        self._chk(self.lib.tg_timer_start(self.h), "tg_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.tg_timer_stop(self.h, C.byref(ms)), "tg_timer_stop")
        return ms.value

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
