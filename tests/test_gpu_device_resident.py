"""Device-resident surfaces: what the engine hands to other GPU code without a trip through host memory, and the
functional batched API whose values stay on the device until they are read.  torch is used HERE (tests only) as the
consumer of `__cuda_array_interface__` / DLPack; nothing in tetris_gymnasium_amd imports it."""
import time

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402
from tetris_gymnasium_amd import engine as engine_mod  # noqa: E402
from tetris_gymnasium_amd.envs import tetris_fn as F  # noqa: E402
from tetris_gymnasium_amd.functional import TETROMINOES, EnvConfig  # noqa: E402
from tetris_gymnasium_amd.vector import TetrisVectorEnv  # noqa: E402

CFG = EnvConfig(width=10, height=20, padding=4, queue_size=7, gravity_enabled=True)


class _D2HCounter:
    """Counts every device → host transfer the Python layer can make (tg_fetch, tg_dev_download, tg_get_state)."""

    def __init__(self, monkeypatch):
        self.n = 0
        for name in ("_fetch", "dev_download", "get_state", "fetch_outputs"):
            orig = getattr(Engine, name)

            def wrapped(eng, *a, _orig=orig, **k):
                self.n += 1
                return _orig(eng, *a, **k)

            monkeypatch.setattr(Engine, name, wrapped)


def _keys(B):
    return np.stack([np.zeros(B, np.uint32), np.arange(1, B + 1, dtype=np.uint32)], axis=1)


def test_batched_step_loop_stays_on_the_device(monkeypatch):
    """1 000 batched_step calls at 65 536 envs: no device → host transfer at all until a leaf is read; then one State costs
    three downloads, an observation one; values equal the oracle's."""
    B, T = 65536, 1000
    keys = _keys(B)
    o = po.OracleBatch(B, 10, 20, po.RULESET_FUNCTIONAL, nthreads=8)
    o.reset((keys[:, 0].astype(np.uint64) << np.uint64(32)) | keys[:, 1].astype(np.uint64))
    F.batched_reset(TETROMINOES, keys, config=CFG)  # (creates the engine)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        new_keys, states, obs = F.batched_reset(TETROMINOES, keys, config=CFG)
        best = min(best, time.perf_counter() - t0)
    assert best < 0.05, f"batched_reset of {B} keys took {best * 1e3:.1f} ms of host time"
    np.testing.assert_array_equal(new_keys[:64], np.array([F.split(k)[0] for k in keys[:64]], np.uint32))
    cnt = _D2HCounter(monkeypatch)
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 7, (T, B)).astype(np.int32)
    t0 = time.perf_counter()
    for t in range(T):
        states, obs, rew, term, info = F.batched_step(TETROMINOES, states, acts[t], config=CFG)
    loop_s = time.perf_counter() - t0
    assert cnt.n == 0, f"{cnt.n} device → host transfers inside the loop"
    assert isinstance(states.board, F.LazyArray) and not states.board.materialized and obs.shape == (B, 20, 10)
    for t in range(T):
        ro, to, lo = o.step(acts[t])
    np.testing.assert_array_equal(np.asarray(rew), ro)
    assert cnt.n == 1  # reward / terminated / lines share one download
    np.testing.assert_array_equal(np.asarray(term), to.astype(bool))
    np.testing.assert_array_equal(np.asarray(info["lines_cleared"]), lo)
    assert cnt.n == 1
    np.testing.assert_array_equal(np.asarray(obs), o.observe(po.OBS_FN))
    assert cnt.n == 2
    bo, reco, rngo = o.get_state()
    np.testing.assert_array_equal(states.board[:, :20, 4:14].astype(np.uint8), bo)
    assert cnt.n == 5  # board + records + keys, once for every leaf of the State
    np.testing.assert_array_equal(np.asarray(states.x), reco["x"]); np.testing.assert_array_equal(np.asarray(states.score), reco["score"])
    np.testing.assert_array_equal(np.asarray(states.rng_key), rngo["key"]); np.testing.assert_array_equal(np.asarray(states.queue), reco["bag"])
    assert cnt.n == 5
    print(f"batched_step loop: {loop_s / T * 1e6:.0f} us of host time per call at {B} envs; batched_reset {best * 1e3:.1f} ms")


def test_old_state_is_restored_device_to_device(monkeypatch):
    """The functions are pure: a State from earlier can be stepped again (device → device restore from the copy it owns,
    no host round trip) and gives what it gave the first time."""
    B = 2048
    keys = _keys(B)
    _, s0, _ = F.batched_reset(TETROMINOES, keys, config=CFG)
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 7, (30, B)).astype(np.int32)
    s = s0
    for t in range(10):
        s, *_ = F.batched_step(TETROMINOES, s, acts[t], config=CFG)
    s10 = s
    firsts = []
    for t in range(10, 20):
        s, obs, rew, term, info = F.batched_step(TETROMINOES, s, acts[t], config=CFG)
        firsts.append((np.asarray(obs), np.asarray(rew), np.asarray(term)))
    cnt = _D2HCounter(monkeypatch)
    uploads = []
    monkeypatch.setattr(Engine, "set_state", lambda *a, **k: uploads.append(1))
    s = s10  # an old State, never read: goes back device → device
    for t in range(10, 20):
        before = cnt.n
        s, obs, rew, term, info = F.batched_step(TETROMINOES, s, acts[t], config=CFG)
        assert cnt.n == before and not uploads  # (the call itself moves nothing to or from the host but the actions)
        for got, exp in zip((np.asarray(obs), np.asarray(rew), np.asarray(term)), firsts[t - 10]):
            np.testing.assert_array_equal(got, exp)
    edited = s10.replace(score=np.asarray(s10.score) + np.float32(5))  # an edited State has no device copy: host upload
    assert edited._token is None and getattr(edited, "_device", None) is None


def test_vector_env_final_observation_info():
    """Same-step autoreset: infos["final_observation"] holds the terminal observation of the envs that ended (gymnasium's
    convention, examples/train_cnn.py:312-316) — equal to what the oracle shows right before it resets them."""
    B = 1500
    v = TetrisVectorEnv(B, observation="board", autoreset_mode="same_step", final_observation=True)
    o = po.OracleBatch(B, 10, 20)
    v.reset(seed=1); o.reset(np.arange(1, B + 1, dtype=np.uint64))
    rng = np.random.default_rng(4)
    seen = 0
    for t in range(150):
        act = rng.integers(0, 8, B)
        obs, rew, term, trunc, infos = v.step(act)
        ro, to, lo = o.step(act.astype(np.int32), autoreset=po.AUTORESET_NONE)
        np.testing.assert_array_equal(term, to.astype(bool))
        if to.any():
            m = to.astype(bool)
            np.testing.assert_array_equal(infos["_final_observation"], m)
            np.testing.assert_array_equal(infos["final_observation"][m], o.observe(po.OBS_BOARD)[m])
            seen += int(m.sum())
            o.reset(mask=to)
        else:
            assert "final_observation" not in infos
        np.testing.assert_array_equal(obs, o.observe(po.OBS_BOARD))
    assert seen > 100
    v.close()


def test_vector_env_four_shards_on_one_device_equal_one_engine():
    """The multi-GPU adapter with every shard on device 0 (this pool has one GPU per box): contiguous shards, global seeds,
    one worker per engine — results independent of the number of shards."""
    B = 4000
    one = TetrisVectorEnv(B, observation="board", autoreset_mode="next_step", devices=(0,))
    four = TetrisVectorEnv(B, observation="board", autoreset_mode="next_step", devices=(0, 0, 0, 0))
    assert [hi - lo for lo, hi in four.ranges] == [1000] * 4 and len(four.engines) == 4
    o1, i1 = one.reset(seed=11); o4, i4 = four.reset(seed=11)
    np.testing.assert_array_equal(o1, o4)
    rng = np.random.default_rng(6)
    for t in range(120):
        act = rng.integers(0, 8, B)
        r1, r4 = one.step(act), four.step(act)
        for x, y in zip(r1[:4], r4[:4]):
            np.testing.assert_array_equal(x, y)
        assert set(r1[4]) == set(r4[4])
        for k in r1[4]:
            if isinstance(r1[4][k], dict):
                for kk in r1[4][k]:
                    np.testing.assert_array_equal(r1[4][k][kk], r4[4][k][kk])
            else:
                np.testing.assert_array_equal(r1[4][k], r4[4][k])
    one.close(); four.close()


def test_vector_env_rollout_device_in_chunks_over_three_shards():
    """TetrisVectorEnv.rollout_device: one process in front of several shards steps them through tg_rollout in chunks (one enqueue per
    shard and 16 transitions instead of one per transition; VERDICT r05 next #5 ii).  Trajectories and final state against the oracle."""
    B, K = 3 * 500, 40
    v = TetrisVectorEnv(B, observation="board", autoreset_mode="same_step", devices=(0, 0, 0))
    v.reset(seed=5)
    o = po.OracleBatch(B)
    o.reset(np.arange(5, 5 + B, dtype=np.uint64))
    acts = np.random.default_rng(3).integers(0, 8, (K, B)).astype(np.int8)
    pa, pobs, pout = [], [], []
    for e, (lo, hi) in zip(v.engines, v.ranges):
        a = np.ascontiguousarray(acts[:, lo:hi])
        d = e.dev_alloc(a.nbytes); e.dev_upload(d, a); pa.append(d)
        pobs.append(e.dev_alloc(K * e.B * 200)); pout.append(e.dev_alloc(K * e.B * N.STEP_OUT.itemsize))
    v.rollout_device(pa, K, pobs, pout, chunk=16)  # 16 + 16 + 8 transitions per shard, nothing waited for
    obs = np.empty((K, B, 20, 10), np.uint8)
    out = np.empty((K, B), N.STEP_OUT)
    for e, (lo, hi), po_, pr_ in zip(v.engines, v.ranges, pobs, pout):
        ob = np.empty((K, e.B, 20, 10), np.uint8); e.dev_download(po_, ob); obs[:, lo:hi] = ob
        ot = np.empty((K, e.B), N.STEP_OUT); e.dev_download(pr_, ot); out[:, lo:hi] = ot
    for k in range(K):
        ro, to, lo_ = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        np.testing.assert_array_equal(out["reward"][k], ro, err_msg=f"reward k={k}")
        np.testing.assert_array_equal(out["terminated"][k], to, err_msg=f"terminated k={k}")
        np.testing.assert_array_equal(out["lines"][k], lo_, err_msg=f"lines k={k}")
        np.testing.assert_array_equal(obs[k], o.observe(po.OBS_BOARD), err_msg=f"observation k={k}")
    with pytest.raises(ValueError):
        v.rollout_device(pa[:2], K)
    for e, *ds in zip(v.engines, pa, pobs, pout):
        for d in ds:
            e.dev_free(d)
    v.close()


def test_engine_knobs_come_from_the_config_not_the_environment(monkeypatch):
    """tg_config.fuse / rng_every / ring_fill_p1 (ABI 2) select the launch shape; results never depend on them."""
    for k in ("TG_FUSE", "TG_RNG_EVERY", "TG_RING_FILL"):
        monkeypatch.delenv(k, raising=False)
    B, K = 1500, 37
    acts = np.random.default_rng(8).integers(0, 8, (K, B)).astype(np.int8)
    ref = None
    for fuse, every, fill in ((0, 0, None), (1, 1, 3), (8, 3, 1), (5, 2, 0), (16, 3, 2), (11, 1, 3)):
        e = Engine(B, 10, 20, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, fuse=fuse, rng_every=every, ring_fill=fill)
        feat = int(e.step_kernel_name(fuse or 16).rstrip(">").split(",")[-1])  # FEAT flags: bit 0 = several transitions per launch
        assert (feat & 1) == (1 if (fuse or 16) > 1 else 0)
        e.reset(np.arange(1, B + 1, dtype=np.uint64))
        res = e.rollout(acts)
        st = e.get_state()
        got = (res[0], res[1], res[2], res[3], st[0], st[1]["bag"], st[2]["state_lo"])
        if ref is None:
            ref = got
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g, r)
        e.close()


@pytest.mark.parametrize("kind", ["board", "dict"])
def test_vector_env_copy_false_returns_pinned_views_with_the_same_content(kind):
    """TetrisVectorEnv(copy=False) = gymnasium's SyncVectorEnv(copy=False): observations are views of two rotating
    page-locked host buffers (tg_host_alloc).  Same content as copy=True at every step; an array stays intact for one
    further step and is reused by the one after."""
    B = 1000
    a = TetrisVectorEnv(B, observation=kind, autoreset_mode="same_step", copy=True)
    b = TetrisVectorEnv(B, observation=kind, autoreset_mode="same_step", copy=False)
    oa, _ = a.reset(seed=3); ob, _ = b.reset(seed=3)
    first = (lambda o: o["board"] if isinstance(o, dict) else o)
    rng = np.random.default_rng(1)
    prev, prev_copy = None, None
    seen_ptrs = set()
    for t in range(6):
        act = rng.integers(0, 8, B)
        oa, ra, ta, _, _ = a.step(act)
        ob, rb, tb, _, _ = b.step(act)
        for k in (oa if isinstance(oa, dict) else [None]):
            np.testing.assert_array_equal(ob[k] if k else ob, oa[k] if k else oa, err_msg=f"obs[{k}] t={t}")
        np.testing.assert_array_equal(rb, ra); np.testing.assert_array_equal(tb, ta)
        if prev is not None:
            np.testing.assert_array_equal(prev, prev_copy, err_msg="the previous step's view was overwritten one step later")
        prev, prev_copy = first(ob), first(ob).copy()
        seen_ptrs.add(first(ob).__array_interface__["data"][0])
    assert len(seen_ptrs) == 2  # two slots, rotating
    a.close(); b.close()


def _addr(a):
    return a.__array_interface__["data"][0]


@pytest.mark.parametrize("kind,shards", [("board", 1), ("dict", 1), ("dict", 2)])
def test_vector_env_copy_true_pool_never_touches_an_array_somebody_holds(kind, shards):
    """copy=True hands out arrays the caller owns.  They come from a pool of page-locked slots that is reused only when nothing
    outside references an array of the slot: env `a` keeps EVERYTHING it was ever given (the pool runs dry, fresh arrays from then
    on), env `b` keeps only a row VIEW of one early observation, env `c` copies at once — all three must tell the same story."""
    import gc

    B = 1000
    mk = lambda: TetrisVectorEnv(B, observation=kind, autoreset_mode="same_step", copy=True, devices=(0,) * shards)
    a, b, c = mk(), mk(), mk()
    first = (lambda o: o["board"] if isinstance(o, dict) else o)
    flat = (lambda o, r, t, info: [o[k] for k in sorted(o)] if isinstance(o, dict) else [o]) 
    kept, truth = [], []
    for env in (a, b, c):
        env.reset(seed=5)
    rng = np.random.default_rng(2)
    b_view = b_view_copy = None
    b_ptrs = []
    for t in range(9):
        act = rng.integers(0, 8, B)
        oa, ra, ta, _, ia = a.step(act)
        kept.append(flat(oa, ra, ta, ia) + [ra, ta, ia["lines_cleared"]])
        ob, rb, tb, _, ib = b.step(act)
        if t == 1:
            b_view = first(ob)[3]  # a view of a view of the slot's block; everything else of this step is dropped
            b_view_copy = b_view.copy()
        b_ptrs.append(_addr(first(ob)))
        del ob, rb, tb, ib
        oc, rc, tc, _, ic = c.step(act)
        truth.append([x.copy() for x in flat(oc, rc, tc, ic) + [rc, tc, ic["lines_cleared"]]])
    for t, (k, tr) in enumerate(zip(kept, truth)):
        for i, (x, y) in enumerate(zip(k, tr)):
            np.testing.assert_array_equal(x, y, err_msg=f"array {i} handed out at step {t} changed while it was held")
    np.testing.assert_array_equal(b_view, b_view_copy, err_msg="a held row view was overwritten")
    if hasattr(__import__("sys"), "getrefcount"):
        host = b._host
        assert host.pool is not None and 2 <= len(host.pool) <= host._POOL_SLOTS
        # b's loop: the slot of step 1 stays busy (the row view), the other slots take turns
        busy = b_ptrs[1]
        assert busy not in b_ptrs[2:], "the slot whose row view is held was written again"
        assert len(set(b_ptrs[2:])) <= host._POOL_SLOTS - 1
        # a's loop kept everything: pool exhausted, fresh arrays after that
        assert len(a._host.pool) == a._host._POOL_SLOTS
        del b_view
        gc.collect()
        ptrs = set()
        for t in range(4):
            ob, *_ = b.step(rng.integers(0, 8, B))
            ptrs.add(_addr(first(ob)))
            del ob
        assert busy in ptrs, "a slot nobody references any more was not taken back into the rotation"
    for env in (a, b, c):
        env.close()


def test_vector_env_copy_fresh_allocates_every_observation():
    """copy="fresh": no pool of page-locked slots — every step returns newly allocated arrays (gymnasium's copy=True to the letter; the
    opt-out ADVICE r05 asked for: the pool's reference-count test cannot see an asynchronous reader that holds no Python reference)."""
    a = TetrisVectorEnv(256, observation="board", autoreset_mode="same_step", copy=True)
    b = TetrisVectorEnv(256, observation="board", autoreset_mode="same_step", copy="fresh")
    oa, _ = a.reset(seed=3); ob, _ = b.reset(seed=3)
    rng = np.random.default_rng(1)
    seen = set()
    for t in range(12):
        act = rng.integers(0, 8, 256)
        ra, rb = a.step(act), b.step(act)
        np.testing.assert_array_equal(ra[0], rb[0]); np.testing.assert_array_equal(ra[1], rb[1]); np.testing.assert_array_equal(ra[2], rb[2])
        keep = rb[0]
        seen.add(keep.ctypes.data)
        del ra, rb
    assert b._host.pool is None and a._host.pool is not None
    with pytest.raises(ValueError):
        TetrisVectorEnv(8, copy="sometimes")
    a.close(); b.close()


def test_lifetimes_after_close_and_unconsumed_dlpack_capsules():
    """(round 3 advisor findings) Page-locked observation arrays outlive `close()` (they own their block, like gymnasium's
    copy=False arrays own theirs); a DLPack capsule that nobody consumes releases what it pinned; a DeviceArray / pooled
    DeviceBuffer of a closed engine fails loudly instead of pointing into freed device memory."""
    import gc

    from tetris_gymnasium_amd import engine as E

    env = TetrisVectorEnv(512, observation="board", autoreset_mode="same_step", copy=False)
    env.reset(seed=2)
    obs, rew, term, _, _ = env.step(np.zeros(512, np.int32))
    keep, ref = obs, obs.copy()
    e = env.engines[0]
    view = e.device_array(N.BUF_OBS)
    n_live = len(E._dl_live)
    cap = view.__dlpack__()
    assert len(E._dl_live) == n_live + 1
    del cap
    gc.collect()
    assert len(E._dl_live) == n_live  # the capsule's destructor ran the deleter
    snap = e.get_state_device()  # pooled device buffers still referenced at close
    env.close()
    gc.collect()
    np.testing.assert_array_equal(keep, ref)  # still readable: the block is freed when `keep` dies
    assert all(b.ptr == 0 for b in snap)
    with pytest.raises(E.EngineError):
        view.__cuda_array_interface__
    del keep
    gc.collect()


def test_set_state_device_is_range_checked_on_the_device():
    """tg_set_state_dev cannot validate device-resident records on the host: a kernel clamps what tg_set_state would refuse (so
    nothing indexes out of range) and the next synchronising call reports it once; valid uploads pass silently."""
    from tetris_gymnasium_amd.engine import EngineError

    B = 256
    e = Engine(B, 10, 20, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
    e.reset(np.arange(1, B + 1, dtype=np.uint64))
    e.step(np.zeros(B, np.int32))
    good = e.get_state_device()
    e.set_state_device(*good)
    e.sync()                                           # nothing to report
    board, rec, rng = e.get_state()
    bad = rec.copy()
    bad["piece"][3] = 9; bad["x"][5] = 200; bad["bag"][7, 2] = 11
    d = e.dev_alloc(bad.nbytes)
    e.dev_upload(d, bad)
    with pytest.raises(EngineError, match="DeviceBuffer"):
        e.set_state_device(rec=d)                      # raw pointers need unchecked=True ...
    with pytest.raises(EngineError, match="board"):
        e.set_state_device(board=good[1])              # ... and a snapshot buffer only fits the slot it was made for (sizes differ)
    with pytest.raises(EngineError, match="rng"):
        e.set_state_device(rng=good[0])
    e.set_state_device(rec=d, unchecked=True)          # ... and are then checked on the device
    with pytest.raises(EngineError, match="tg_set_state_dev"):
        e.sync()
    e.sync()                                           # reported once
    _, rec2, _ = e.get_state()
    assert rec2["piece"][3] <= 6 and rec2["x"][5] <= 14 and rec2["bag"][7, 2] <= 6
    for _ in range(20):                                # the engine keeps running on the clamped state
        e.step(np.random.default_rng(0).integers(0, 8, B))
    e.dev_free(d)
    e.close()
