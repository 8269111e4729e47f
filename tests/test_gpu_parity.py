"""GPU parity tests proper: the HIP engine (through the C ABI) against (a) golden vectors generated from
the reference NumPy env and (b) the CPU oracle on identical seeds and actions.  Bit-exact: boards,
rewards, terminated, lines, observations, piece/queue/bag/RNG state."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay

pytestmark = pytest.mark.gpu

from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402


class EngineBackend:
    def __init__(self, batch, width, height, gravity, queue_size):
        self.e = Engine(batch, width, height, ruleset=N.RULESET_NUMPY, gravity=gravity, queue_size=queue_size,
                        obs_mode=N.OBS_DICT)

    def reset(self, seeds=None, mask=None):
        self.e.reset(seeds, mask)

    def step(self, actions):
        return self.e.step(actions)

    def observe_dict(self):
        return self.e.observe()

    def get_state(self):
        return self.e.get_state()

    def set_state(self, **kw):
        self.e.set_state(**kw)


class EngineInStepObsBackend(EngineBackend):
    """The same, but the dict observation is what the STEP KERNEL packed (TG_BUF_OBS as the transition left it: board and
    mask arrays written by its two waves, holder / queue strips stored only when they change) — the surface that is
    benchmarked — instead of a re-pack by the observe kernel.  After a reset / set_state the engine re-packs anyway."""

    def observe_dict(self):
        return self.e.observation()


TRACES = ["trace_10x20_g1", "trace_10x20_g0", "trace_20x40_g1", "trace_6x12_g1", "clears_10x20", "clears_20x40",
          "trace_7x14_q2_g1", "trace_12x24_q7_g0"]  # (the last two: board sizes the engine compiles at run time)


@pytest.mark.parametrize("name", TRACES)
def test_engine_replays_reference_trace(golden_dir, name):
    replay.replay_trace(EngineBackend, replay.load(golden_dir, name))


@pytest.mark.parametrize("name", TRACES)
def test_engine_replays_reference_trace_with_the_in_step_observation(golden_dir, name):
    """Every reference-generated observation checksum of the traces against the observation the step kernel itself packs."""
    be = replay.replay_trace(EngineInStepObsBackend, replay.load(golden_dir, name))
    # (the step kernel packs the dict surface itself where the geometry allows — BitsGeo::DICT_OK: even width and height, padded board a
    # multiple of 16 bytes, a whole number of 16-env groups per workgroup; the other sizes go through the observe kernel)
    assert be.e.dict_in_kernel() or name.startswith(("trace_6x12", "trace_7x14"))


def _cmp_state(e, o, msg=""):
    be, re_, ge = e.get_state()
    bo, ro, go = o.get_state()
    np.testing.assert_array_equal(be, bo, err_msg="board " + msg)
    for k in re_.dtype.names:
        if k == "reserved":
            continue
        np.testing.assert_array_equal(re_[k], ro[k], err_msg=f"rec.{k} {msg}")
    return ge, go


@pytest.mark.parametrize("W,H,Q,gravity", [(10, 20, 4, True), (10, 20, 4, False), (20, 40, 7, True), (8, 16, 2, True),
                                            (12, 24, 4, True), (10, 24, 1, True)])
@pytest.mark.parametrize("autoreset", [N.AUTORESET_NONE, N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
def test_numpy_rules_engine_vs_oracle(W, H, Q, gravity, autoreset):
    B, T = 512, 300
    e = Engine(B, W, H, ruleset=N.RULESET_NUMPY, gravity=gravity, queue_size=Q, obs_mode=N.OBS_BOARD,
               autoreset=autoreset)
    o = po.OracleBatch(B, W, H, po.RULESET_NUMPY, gravity, Q)
    seeds = np.arange(1000, 1000 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD))
    rng = np.random.default_rng(W * 100 + H + autoreset)
    done_total = 0
    for t in range(T):
        a = rng.integers(0, 8, B)
        re_, te, le = e.step(a)
        ro, to, lo = o.step(a, autoreset=autoreset)
        np.testing.assert_array_equal(re_, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(te, to, err_msg=f"terminated t={t}")
        np.testing.assert_array_equal(le, lo, err_msg=f"lines t={t}")
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD), err_msg=f"obs t={t}")
        done_total += int(te.sum())
        if autoreset == N.AUTORESET_NONE and te.any():
            e.reset(mask=te); o.reset(mask=te)
        if t % 25 == 0 or t == T - 1:
            ge, go = _cmp_state(e, o, f"t={t}")
            for k in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
                np.testing.assert_array_equal(ge[k], go[k], err_msg=f"rng.{k} t={t}")
    assert done_total > B  # several episodes per env ended (reset/autoreset path exercised)


@pytest.mark.parametrize("B", [1, 31, 32, 33, 36, 60, 63, 64, 65, 100, 128 + 40])
@pytest.mark.parametrize("ruleset", [N.RULESET_NUMPY, N.RULESET_FUNCTIONAL])
def test_ragged_batches_one_and_eight_transitions_per_launch(B, ruleset, monkeypatch):
    """The last workgroup of a batch that is no multiple of 64: dead lanes share the LDS slot of the last live env, and with
    two stream waves per workgroup (the instantiation small NUMPY batches pick) the second wave owns envs 32..63 of the
    workgroup — none (B % 64 <= 32: it only passes the barriers), some, or all of them.  One transition per launch and the
    fused rollout, both against the oracle; TG_SPLIT=0 (one stream wave) must give the same."""
    fn = ruleset == N.RULESET_FUNCTIONAL
    obs, po_obs, na = (N.OBS_FN, po.OBS_FN, 7) if fn else (N.OBS_BOARD, po.OBS_BOARD, 8)
    T = 160
    acts = np.random.default_rng(B).integers(0, na, (T, B)).astype(np.int8)
    seeds = np.arange(7, 7 + B, dtype=np.uint64)
    o = po.OracleBatch(B, 10, 20, ruleset)
    o.reset(seeds)
    exp = []
    for t in range(T):
        r, d, l = o.step(acts[t].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        exp.append((r.copy(), d.copy(), l.copy(), o.observe(po_obs).copy()))
    e = Engine(B, 10, 20, ruleset=ruleset, obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP)
    if not fn and os.environ.get("TG_SPLIT") != "0":
        assert int(e.step_kernel_name().rstrip(">").split(",")[-1]) & 128  # two stream waves per workgroup
    e.reset(seeds)
    for t in range(T):
        r, d, l = e.step(acts[t])
        for g, x, name in zip((r, d, l, e.observation()), exp[t], ("reward", "terminated", "lines", "obs")):
            np.testing.assert_array_equal(g, x, err_msg=f"{name} t={t}")
    _cmp_state(e, o, "one transition per launch")
    if B % 4:  # (tg_rollout packs int8 action rows: batch must be a multiple of 4)
        e.close()
        return
    e.reset(seeds)
    res = e.rollout(acts)  # 8 transitions per launch, trajectory outputs
    for t in range(T):
        np.testing.assert_array_equal(res[0][t], exp[t][3], err_msg=f"rollout obs t={t}")
        np.testing.assert_array_equal(res[1][t], exp[t][0], err_msg=f"rollout reward t={t}")
    _cmp_state(e, o, "fused rollout")
    e.close()


@pytest.mark.parametrize("W,H,gravity", [(10, 20, True), (10, 20, False), (20, 40, True), (6, 12, True)])
@pytest.mark.parametrize("autoreset", [N.AUTORESET_NONE, N.AUTORESET_SAME_STEP])
def test_functional_rules_engine_vs_oracle(W, H, gravity, autoreset):
    B, T = 512, 300
    e = Engine(B, W, H, ruleset=N.RULESET_FUNCTIONAL, gravity=gravity, obs_mode=N.OBS_FN, autoreset=autoreset)
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, gravity)
    keys = (np.arange(B, dtype=np.uint64) * 7919 + 42)
    e.reset(keys); o.reset(keys)
    np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN))
    rng = np.random.default_rng(5 + W + autoreset)
    done_total = 0
    for t in range(T):
        a = rng.integers(-1, 9, B) if t % 10 == 0 else rng.integers(0, 7, B)  # incl. ids lax.switch clamps
        re_, te, le = e.step(a)
        ro, to, lo = o.step(a, autoreset=autoreset)
        np.testing.assert_array_equal(re_, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(te, to, err_msg=f"terminated t={t}")
        np.testing.assert_array_equal(le, lo, err_msg=f"lines t={t}")
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN), err_msg=f"obs t={t}")
        done_total += int(te.sum())
        if autoreset == N.AUTORESET_NONE and te.all():
            e.reset(mask=te); o.reset(mask=te)  # reset(rng_key) continuation
        if t % 25 == 0 or t == T - 1:
            ge, go = _cmp_state(e, o, f"t={t}")
            np.testing.assert_array_equal(ge["key"], go["key"], err_msg=f"rng_key t={t}")
    assert done_total > 0


def test_line_clear_dense_boards_engine_vs_oracle():
    """Constructed boards (the reference's own test style): every lock completes rows."""
    B, W, H = 256, 10, 20
    rng = np.random.default_rng(11)
    for ruleset, obs in ((N.RULESET_NUMPY, N.OBS_BOARD), (N.RULESET_FUNCTIONAL, N.OBS_FN)):
        e = Engine(B, W, H, ruleset=ruleset, obs_mode=obs)
        o = po.OracleBatch(B, W, H, ruleset)
        seeds = np.arange(1, B + 1, dtype=np.uint64)
        e.reset(seeds); o.reset(seeds)
        total = 0
        for rep in range(40):
            b = rng.integers(2, 9, (B, H, W)).astype(np.uint8)
            top = rng.integers(0, 14, B)
            for i in range(B):
                b[i, : top[i]] = 0
                g = rng.integers(0, W - 3)
                b[i, :, g:g + rng.integers(1, 5)] = 0
                if rep % 4 == 0:
                    b[i, 0, :] = rng.integers(0, 9, W)  # non-empty row 0: the jnp.take wrap corner
            be, rc, _ = e.get_state()
            rc["flags"] = 0
            rc["x"] = rng.integers(3, 13, B); rc["y"] = 0; rc["rot"] = rng.integers(0, 4, B)
            e.set_state(board=b, rec=rc); o.set_state(board=b, rec=rc)
            drop = 5 if ruleset == N.RULESET_NUMPY else 6
            for _ in range(3):
                a = np.full(B, drop)
                re_, te, le = e.step(a)
                ro, to, lo = o.step(a)
                np.testing.assert_array_equal(re_, ro); np.testing.assert_array_equal(te, to)
                np.testing.assert_array_equal(le, lo)
                total += int(le.sum())
                _cmp_state(e, o, f"rep={rep}")
        assert total > 300


def test_swap_and_hold_paths():
    B = 256
    e = Engine(B, obs_mode=N.OBS_DICT)
    o = po.OracleBatch(B)
    seeds = np.arange(50, 50 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(2)
    for t in range(400):
        a = rng.choice([5, 6, 6, 7, 0, 1, 3], B)
        re_, te, le = e.step(a)
        ro, to, lo = o.step(a)
        np.testing.assert_array_equal(re_, ro); np.testing.assert_array_equal(te, to)
        od = o.observe(po.OBS_DICT)
        ed = e.observation()
        for k in od:
            np.testing.assert_array_equal(ed[k], od[k], err_msg=f"{k} t={t}")
        if te.any():
            e.reset(mask=te); o.reset(mask=te)
    _cmp_state(e, o)


def test_post_game_over_stepping_matches_reference_quirk():
    """NUMPY rules keep stepping after game over (SURVEY §8.0 'After game over')."""
    B = 64
    e = Engine(B, obs_mode=N.OBS_BOARD)
    o = po.OracleBatch(B)
    e.reset(np.arange(1, B + 1, dtype=np.uint64)); o.reset(np.arange(1, B + 1, dtype=np.uint64))
    rng = np.random.default_rng(9)
    for t in range(600):
        a = rng.integers(0, 8, B)
        assert all(np.array_equal(x, y) for x, y in zip(e.step(a), o.step(a)))
    _cmp_state(e, o)
    assert (e.get_state()[1]["flags"] & 1).sum() > 0


def test_invalid_action_is_rejected_host_side():
    e = Engine(8)
    e.reset(np.arange(1, 9, dtype=np.uint64))
    from tetris_gymnasium_amd import EngineError
    with pytest.raises(EngineError, match="outside Discrete"):
        e.step(np.array([0, 1, 2, 3, 4, 5, 6, 8]))


def test_step_k_equals_k_steps_and_episode_stats():
    B, K = 2048, 64
    rng = np.random.default_rng(4)
    acts = rng.integers(0, 8, (K, B)).astype(np.int8)
    seeds = np.arange(1, B + 1, dtype=np.uint64)
    e1 = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, episode_stats=True)
    e2 = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, episode_stats=True)
    e1.reset(seeds); e2.reset(seeds)
    ret = np.zeros(B)
    for k in range(K):
        r, t, l = e1.step(acts[k].astype(np.int32))
    d = e2.dev_alloc(acts.nbytes)
    e2.dev_upload(d, acts)
    e2.step_k_device(d, K)
    assert all(np.array_equal(x, y) for x, y in zip(e1.fetch_outputs(), e2.fetch_outputs()))
    np.testing.assert_array_equal(e1.observation(), e2.observation())
    s1, s2 = e1.get_state(), e2.get_state()
    assert all(a.tobytes() == b.tobytes() for a, b in zip(s1, s2))
    np.testing.assert_array_equal(e1.episode_stats(), e2.episode_stats())
    assert e1.episode_stats()["last_len"].max() > 0
    e2.dev_free(d)


def test_full_size_determinism_and_invariants():
    """BASELINE size (65 536 envs): size-independent properties — two identical runs agree bit-for-bit,
    cell counts are consistent with locks/clears, boards never hold a full row, sharding is transparent."""
    B, T = 65536, 48
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 8, (T, B)).astype(np.int32)
    seeds = np.arange(1, B + 1, dtype=np.uint64)
    runs = []
    for rep in range(2):
        e = Engine(B, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
        e.reset(seeds)
        tot_r = np.zeros(B)
        for t in range(T):
            r, te, l = e.step(acts[t])
            tot_r += r
        board, rec, g = e.get_state()
        runs.append((board.copy(), rec.copy(), g.copy(), tot_r, e.observation().copy()))
        e.close()
    for a, b in zip(runs[0], runs[1]):
        assert a.tobytes() == b.tobytes()
    board, rec = runs[0][0], runs[0][1]
    assert not (board != 0).all(axis=2).any(), "a completed row survived"
    assert ((board == 0) | ((board >= 2) & (board <= 8))).all()
    assert (board != 0).sum(axis=(1, 2)).max() % 1 == 0 and ((board != 0).sum(axis=(1, 2)) % 2 == 0).all()  # 4 cells per lock, 10 per clear
    # sharding transparency: two half-size engines with env_offset reproduce the full batch
    halves = []
    for k in range(2):
        sl = slice(k * B // 2, (k + 1) * B // 2)
        e = Engine(B // 2, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, env_offset=k * B // 2)
        e.reset(seeds[sl])
        for t in range(T):
            e.step_async(acts[t][sl])
        halves.append(e.get_state()[0])
        e.close()
    np.testing.assert_array_equal(np.concatenate(halves), board)
    # a 256-env sample against the oracle at this size
    idx = rng.choice(B, 256, replace=False)
    o = po.OracleBatch(256)
    o.reset(seeds[idx])
    for t in range(T):
        o.step(acts[t][idx], autoreset=po.AUTORESET_SAME_STEP)
    np.testing.assert_array_equal(o.get_state()[0], board[idx])


@pytest.mark.parametrize("ruleset,B", [(N.RULESET_NUMPY, 20000), (N.RULESET_NUMPY, 33000), (N.RULESET_FUNCTIONAL, 20000)])
def test_sharded_streams_engine_vs_oracle(ruleset, B, monkeypatch):
    """Batches large enough for the engine to spread a transition over several streams (contiguous env shards, one
    launch each, ragged last shard) must give exactly the single-launch result: host actions, device int8 action
    rows through tg_step_k, masked resets in between, state and generator compared at the end."""
    W, H, T = 10, 20, 60
    numpy_rules = ruleset == N.RULESET_NUMPY
    obs = N.OBS_BOARD if numpy_rules else N.OBS_FN
    po_obs = po.OBS_BOARD if numpy_rules else po.OBS_FN
    na = 8 if numpy_rules else 7
    e = Engine(B, W, H, ruleset=ruleset, obs_mode=obs, autoreset=N.AUTORESET_NONE, episode_stats=True, streams=4)  # (tg_config.streams; off by default: see DESIGN.md)
    o = po.OracleBatch(B, W, H, ruleset)
    seeds = np.arange(5, 5 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(B)
    for t in range(T):
        a = rng.integers(0, na, B)
        re_, te, le = e.step(a)
        ro, to, lo = o.step(a, autoreset=po.AUTORESET_NONE)
        np.testing.assert_array_equal(re_, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(te, to, err_msg=f"terminated t={t}")
        np.testing.assert_array_equal(le, lo, err_msg=f"lines t={t}")
        if t % 10 == 0:
            np.testing.assert_array_equal(e.observation(), o.observe(po_obs), err_msg=f"obs t={t}")
        if te.any():
            e.reset(mask=te); o.reset(mask=te)
    # device-resident int8 action rows (the bench path): K launches per shard without host round trips
    K = 24
    acts = rng.integers(0, na, (K, B)).astype(np.int8)
    # (no env may terminate un-reset in a rollout without autoreset: NUMPY rules keep stepping a dead env, as the oracle)
    d = e.dev_alloc(acts.nbytes)
    e.dev_upload(d, acts)
    e.step_k_device(d, K)
    for k in range(K):
        ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_NONE)
    np.testing.assert_array_equal(e.observation(), o.observe(po_obs))
    got = e.fetch_outputs()
    np.testing.assert_array_equal(got[0], ro); np.testing.assert_array_equal(got[1], to); np.testing.assert_array_equal(got[2], lo)
    ge, go = _cmp_state(e, o, "end")
    if numpy_rules:
        for k in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
            np.testing.assert_array_equal(ge[k], go[k], err_msg=f"rng.{k}")
    e.dev_free(d)
    e.close()


@pytest.mark.parametrize("fuse,ring_fill", [(1, 4), (2, 4), (4, 4), (8, 4), (8, 1), (3, 0), (16, 4), (16, 1), (11, 4)])
@pytest.mark.parametrize("ruleset,W,H,Q,B,autoreset,grouped", [
    (N.RULESET_NUMPY, 10, 20, 4, 1000, N.AUTORESET_SAME_STEP, False),
    (N.RULESET_NUMPY, 10, 20, 4, 4096, N.AUTORESET_NEXT_STEP, False),
    (N.RULESET_NUMPY, 10, 20, 7, 700, N.AUTORESET_SAME_STEP, False),   # queue 7: resets consume two shuffles
    (N.RULESET_NUMPY, 20, 40, 7, 300, N.AUTORESET_SAME_STEP, True),    # grouped hard drops: a lock every transition
    (N.RULESET_NUMPY, 6, 12, 2, 500, N.AUTORESET_NONE, False),
    (N.RULESET_FUNCTIONAL, 10, 20, 7, 1000, N.AUTORESET_SAME_STEP, False),
])
def test_fused_rollout_equals_single_transitions(fuse, ring_fill, ruleset, W, H, Q, B, autoreset, grouped, monkeypatch):
    """tg_step_k keeps state on chip across up to TG_FUSE transitions per launch (one barrier per transition, the
    observation of every transition still written): final observation, outputs, state, generator and episode
    statistics must equal K single transitions of the oracle, for K that are and are not multiples of the group."""
    monkeypatch.setenv("TG_FUSE", str(fuse))
    # ring_fill < 4: the RNG workgroups keep fewer look-ahead permutations, so the logic wave's in-line shuffle (ring run
    # dry: practically unreachable with the full ring) is exercised all the time; 0 = it does every shuffle itself
    monkeypatch.setenv("TG_RING_FILL", str(ring_fill))
    numpy_rules = ruleset == N.RULESET_NUMPY
    obs = N.OBS_BOARD if numpy_rules else N.OBS_FN
    po_obs = po.OBS_BOARD if numpy_rules else po.OBS_FN
    na = 4 * W if grouped else (8 if numpy_rules else 7)
    e = Engine(B, W, H, ruleset=ruleset, queue_size=Q, obs_mode=obs, autoreset=autoreset, grouped=grouped, episode_stats=True)
    o = po.OracleBatch(B, W, H, ruleset, True, Q)
    seeds = np.arange(77, 77 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(fuse * 100 + W)
    ret = np.zeros(B); ln = np.zeros(B, np.int64)
    for K in (1, 2, 3, 5, 8, 13, 40, 64):
        acts = rng.integers(0, na, (K, B)).astype(np.int8)
        d = e.dev_alloc(acts.nbytes)
        e.dev_upload(d, acts)
        e.step_k_device(d, K)
        for k in range(K):
            if grouped:
                ro, to, lo, _ = o.grouped_step(acts[k].astype(np.int32), autoreset=autoreset)
            else:
                ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=autoreset)
        got = e.fetch_outputs()
        np.testing.assert_array_equal(got[0], ro, err_msg=f"reward K={K}")
        np.testing.assert_array_equal(got[1], to, err_msg=f"terminated K={K}")
        np.testing.assert_array_equal(got[2], lo, err_msg=f"lines K={K}")
        np.testing.assert_array_equal(e.observation(), o.observe(po_obs), err_msg=f"obs K={K}")
        ge, go = _cmp_state(e, o, f"K={K}")
        if numpy_rules:
            for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
                np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f} K={K}")
        e.dev_free(d)
        if autoreset == N.AUTORESET_NONE:
            _, term, _ = got
            if term.any():
                e.reset(mask=term); o.reset(mask=term)
    e.close()


@pytest.mark.parametrize("every", [1, 2, 3])
@pytest.mark.parametrize("W,H,Q,B,grouped", [(10, 20, 4, 1000, False), (10, 20, 7, 600, True), (6, 12, 2, 500, True),
                                             (10, 20, 4, 132, False), (10, 20, 4, 64, False)])
def test_rng_role_every_nth_launch(every, W, H, Q, B, grouped, monkeypatch):
    """One-transition launches carry the RNG-role workgroups only every TG_RNG_EVERY-th time (the ring parity then flips
    less often than the consumed-counter parity); whatever the cadence — and with single steps, rollouts, state
    round trips and masked resets in between, batches of 1 000 / 132 / 64 envs — pieces and generator must equal the oracle's."""
    monkeypatch.setenv("TG_RNG_EVERY", str(every))
    monkeypatch.setenv("TG_FUSE", "1")
    na = 4 * W if grouped else 8
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, grouped=grouped)
    o = po.OracleBatch(B, W, H, N.RULESET_NUMPY, True, Q)
    seeds = np.arange(5, 5 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(every * 10 + W)

    def ostep(a):
        return o.grouped_step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)[:3] if grouped \
            else o.step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)

    for rnd in range(6):
        for _ in range(1 + rnd):  # an odd / even number of single launches between the other entry points
            a = rng.integers(0, na, B).astype(np.int8)
            got = e.step(a)
            ro, to, lo = ostep(a)
            np.testing.assert_array_equal(got[0], ro); np.testing.assert_array_equal(got[1], to)
        K = (5, 8, 13)[rnd % 3]
        acts = rng.integers(0, na, (K, B)).astype(np.int8)
        d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts); e.step_k_device(d, K); e.dev_free(d)
        for k in range(K):
            ostep(acts[k])
        if rnd % 2:
            e.set_state(*e.get_state())
        else:
            m = rng.random(B) < 0.2
            e.reset(mask=m); o.reset(mask=m)
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD), err_msg=f"obs round {rnd}")
        ge, go = _cmp_state(e, o, f"round {rnd}")
        for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
            np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f} round {rnd}")
    e.close()


@pytest.mark.parametrize("fuse", [1, 16])
@pytest.mark.parametrize("W,H,Q,B,grouped", [(10, 20, 4, 1000, False), (20, 40, 7, 300, True), (6, 12, 2, 100, False), (12, 24, 4, 77 * 4, False),
                                             (20, 40, 7, 32 * 9 + 20, True), (20, 40, 4, 32 * 5 + 16, False),  # (kFeatSeq: ragged second / empty second group)
                                             (22, 30, 4, 32 * 2 + 24, False),   # (... of a board size compiled at run time: 660 cells, 64-bit column masks)
                                             (32, 24, 4, 32 + 28, False)])       # (... the widest board: 768 cells, 32-bit masks, every row-mask bit in use)
def test_packed_board_handles(fuse, W, H, Q, B, grouped, monkeypatch):
    """TG_PACKED=1: the handle keeps its board as nibbles (tg_kernel_bits.h kFeatPacked — what large working sets pick by
    themselves) and the uint8 TG_BUF_BOARD is brought up to date on demand.  Single steps, rollouts, trajectories, masked
    resets, state round trips (host and device), board fetches and ragged last workgroups against the oracle; cell values
    the nibbles cannot hold are refused by tg_set_state."""
    monkeypatch.setenv("TG_PACKED", "1")
    monkeypatch.setenv("TG_FUSE", str(fuse))
    na = 4 * W if grouped else 8
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, grouped=grouped)
    assert int(e.step_kernel_name().rstrip(">").split(",")[-1]) & 256
    # (boards above 640 cells: a one-transition launch is the 32-env workgroup whose stream wave walks two slab groups — kFeatSeq)
    assert bool(int(e.step_kernel_name(1).rstrip(">").split(",")[-1]) & 2048) == (W * H > 640)
    o = po.OracleBatch(B, W, H, N.RULESET_NUMPY, True, Q)
    seeds = np.arange(5, 5 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(fuse * 10 + W)

    def ostep(a):
        return o.grouped_step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)[:3] if grouped \
            else o.step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)

    for rnd in range(6):
        for _ in range(1 + rnd):
            a = rng.integers(0, na, B).astype(np.int8)
            if not grouped:
                a[rng.random(B) < 0.3] = 5
            got = e.step(a)
            ro, to, lo = ostep(a)
            np.testing.assert_array_equal(got[0], ro); np.testing.assert_array_equal(got[1], to)
            np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD))
        K = (5, 16, 13)[rnd % 3]
        acts = rng.integers(0, na, (K, B)).astype(np.int8)
        if rnd == 2:  # with the trajectory
            res = e.rollout(acts)
            for k in range(K):
                ro, to, lo = ostep(acts[k])
                np.testing.assert_array_equal(res[0][k], o.observe(po.OBS_BOARD), err_msg=f"traj obs k={k}")
                np.testing.assert_array_equal(res[1][k], ro)
        else:
            d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts); e.step_k_device(d, K); e.dev_free(d)
            for k in range(K):
                ostep(acts[k])
        if rnd % 3 == 1:
            e.set_state(*e.get_state())                      # host round trip
        elif rnd % 3 == 2:
            e.set_state_device(*e.get_state_device())        # device round trip
        else:
            m = rng.random(B) < 0.2
            e.reset(mask=m); o.reset(mask=m)                 # the other envs' boards must survive
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD), err_msg=f"obs round {rnd}")
        ge, go = _cmp_state(e, o, f"round {rnd}")           # (boards through tg_get_state)
        for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
            np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f} round {rnd}")
        np.testing.assert_array_equal(e.device_array(N.BUF_BOARD).numpy(), o.get_state()[0])
    b, r, g = e.get_state()
    b[0, 0, 0] = 16
    with pytest.raises(Exception, match="nibbles"):
        e.set_state(board=b, rec=r, rng=g)
    e.close()


@pytest.mark.parametrize("fuse", [1, 16])
@pytest.mark.parametrize("autoreset", [N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
@pytest.mark.parametrize("W,H,B", [(10, 20, 1000), (12, 24, 77 * 4), (6, 12, 100), (20, 40, 32 * 4 + 20), (22, 30, 32 + 12)])
def test_packed_board_functional_handles(fuse, autoreset, W, H, B, monkeypatch):
    """TG_PACKED=1 with the FUNCTIONAL rule-set (what its handles above ~50 Mi cells pick by themselves): single steps,
    rollouts with and without the trajectory, masked resets, host and device state round trips, uploaded tall boards and
    game-over states against the oracle, jax keys included.  A lock there is `board += id` (functional/core.py:103-121): in
    play that stays a nibble; a piece uploaded INSIDE the stack makes the sum leave the range, which the step kernel notices
    and the next wait reports (TG_EINVAL) instead of keeping a wrong board silently."""
    monkeypatch.setenv("TG_PACKED", "1")
    monkeypatch.setenv("TG_FUSE", str(fuse))
    e = Engine(B, W, H, ruleset=N.RULESET_FUNCTIONAL, queue_size=7, obs_mode=N.OBS_FN, autoreset=autoreset)
    assert int(e.step_kernel_name().rstrip(">").split(",")[-1]) & 256
    assert bool(int(e.step_kernel_name(1).rstrip(">").split(",")[-1]) & 2048) == (W * H > 640)  # (kFeatSeq)
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, True, 7)
    seeds = np.arange(31, 31 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(fuse * 100 + W + autoreset)
    n_done = 0

    def check(msg):
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN), err_msg="obs " + msg)
        ge, go = _cmp_state(e, o, msg)  # (boards through tg_get_state: unpacked on demand)
        np.testing.assert_array_equal(ge["key"], go["key"], err_msg="rng.key " + msg)
        np.testing.assert_array_equal(e.device_array(N.BUF_BOARD).numpy(), o.get_state()[0], err_msg="TG_BUF_BOARD " + msg)

    for rnd in range(6):
        if rnd == 3:  # tall stacks (ids 2..8), one free column: locks, line clears (vacated rows = copies of old row 0) and game overs
            b, r, g = o.get_state()
            for i in range(B):
                top = 3 + 2 * (i % 3)
                b[i, top:, :] = 2 + (i % 7)
                b[i, top:, i % W] = 0
            r["y"][:] = 0; r["rot"][:] = 0; r["x"][:] = (W + 8) // 2 - 2  # (every piece back at its spawn cell, above the stacks)
            o.set_state(board=b, rec=r, rng=g); e.set_state(board=b, rec=r, rng=g)
        if rnd == 5:  # every fifth env is handed over already game over
            b, r, g = o.get_state()
            r["flags"][::5] |= N.F_GAME_OVER
            o.set_state(board=b, rec=r, rng=g); e.set_state(board=b, rec=r, rng=g)
        for _ in range(1 + rnd):
            a = rng.integers(0, 7, B).astype(np.int8)
            a[rng.random(B) < 0.4] = 6
            got = e.step(a)
            exp = o.step(a.astype(np.int32), autoreset=autoreset)
            for g_, x, nm in zip(got, exp, ("reward", "terminated", "lines")):
                np.testing.assert_array_equal(g_, x, err_msg=f"{nm} round {rnd}")
            n_done += int(exp[1].sum())
            np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN))
        K = (5, 16, 13)[rnd % 3]
        acts = rng.integers(0, 7, (K, B)).astype(np.int8)
        acts[rng.random((K, B)) < 0.4] = 6
        if rnd == 2:  # with the trajectory
            res = e.rollout(acts)
            for k in range(K):
                ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=autoreset)
                n_done += int(to.sum())
                np.testing.assert_array_equal(res[0][k], o.observe(po.OBS_FN), err_msg=f"traj obs k={k}")
                np.testing.assert_array_equal(res[1][k], ro)
        else:
            d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts); e.step_k_device(d, K); e.dev_free(d)
            for k in range(K):
                n_done += int(o.step(acts[k].astype(np.int32), autoreset=autoreset)[1].sum())
        if rnd % 3 == 1:
            e.set_state(*e.get_state())                      # host round trip
        elif rnd % 3 == 2:
            e.set_state_device(*e.get_state_device())        # device round trip
        else:
            m = rng.random(B) < 0.2
            ks = rng.integers(1, 1 << 40, B).astype(np.uint64)
            e.reset(ks, mask=m); o.reset(ks, mask=m)         # the other envs' boards must survive
        check(f"round {rnd}")
    assert n_done > B
    # cell values the nibbles cannot hold are refused outright ...
    b, r, g = e.get_state()
    b2 = b.copy(); b2[0, 0, 0] = 16
    with pytest.raises(Exception, match="nibbles"):
        e.set_state(board=b2, rec=r, rng=g)
    # ... and a piece uploaded INSIDE a stack of 15s (its lock adds 2..8 onto them) is reported by the wait behind that lock
    b[:, 4:, :] = 15
    r["flags"][:] = 0
    r["y"][:] = 6; r["x"][:] = 6; r["rot"][:] = 0
    e.set_state(board=b, rec=r, rng=g)
    with pytest.raises(Exception, match="nibble range"):
        e.step(np.full(B, 6, np.int8))
        e.sync()
    e.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_packed_functional_exact_path_reads_the_current_board(overlap, monkeypatch):
    """behind a tg_set_state_dev a FUNCTIONAL handle takes the exact path (one transition per launch + the observe
    kernel) until something waits on the stream, and for good while an uploaded piece lies on filled cells.  On a nibble-packed handle the
    step kernel maintains the nibbles only, so the observe kernel must see the uint8 board brought up to date FIRST — it used to re-pack
    the boards of the last upload.  No sync between the upload and the transitions; hard drops so that every transition locks."""
    monkeypatch.setenv("TG_PACKED", "1")
    B, W, H = 700, 10, 20
    e = Engine(B, W, H, ruleset=N.RULESET_FUNCTIONAL, queue_size=7, obs_mode=N.OBS_FN, autoreset=N.AUTORESET_SAME_STEP)
    assert int(e.step_kernel_name().rstrip(">").split(",")[-1]) & 256
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, True, 7)
    seeds = np.arange(77, 77 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(5 + overlap)
    for _ in range(6):
        a = rng.integers(0, 7, B).astype(np.int8)
        e.step(a); o.step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
    if overlap:  # a low stack of 2s under every second env's piece: its lock adds 2..8 onto them — sums that still fit a nibble
        b, r, g = o.get_state()
        b[::2, 1:3, :] = 2
        r["y"][::2] = 0; r["rot"][::2] = 0; r["x"][::2] = (W + 8) // 2 - 2
        r["flags"][::2] = 0
        o.set_state(board=b, rec=r, rng=g); e.set_state(board=b, rec=r, rng=g)
        e.set_state_device(*e.get_state_device())
    else:
        e.set_state_device(*e.get_state_device())  # (no wait behind it: fn_overlap_pending)
    for k in range(5):
        a = np.full(B, 6, np.int8) if k % 2 == 0 else rng.integers(0, 7, B).astype(np.int8)
        if k == 3:  # a fused request in the same window: still the exact path underneath
            acts = np.stack([a, np.full(B, 6, np.int8), a])
            d = e.dev_alloc(acts.nbytes); e.dev_upload(d, acts); e.step_k_device(d, 3); e.dev_free(d)
            for row in acts:
                o.step(row.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        else:
            got = e.step(a)
            exp = o.step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            np.testing.assert_array_equal(got[0], exp[0]); np.testing.assert_array_equal(got[1], exp[1])
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN), err_msg=f"obs after transition {k}")
    _cmp_state(e, o, "end")
    e.close()


@pytest.mark.parametrize("fuse", [1, 8, 16])
@pytest.mark.parametrize("ruleset,obs,B", [(N.RULESET_NUMPY, N.OBS_BOARD, 1500), (N.RULESET_FUNCTIONAL, N.OBS_FN, 640),
                                           (N.RULESET_NUMPY, N.OBS_RGB, 200)])
def test_rollout_trajectory_equals_stepwise(fuse, ruleset, obs, B, monkeypatch):
    """tg_rollout keeps the whole trajectory (observation + reward / terminated / lines of every transition): must
    equal stepping the oracle one transition at a time — written by the fused kernel itself (BOARD / FN) or by the
    copy fallback (RGB)."""
    monkeypatch.setenv("TG_FUSE", str(fuse))
    numpy_rules = ruleset == N.RULESET_NUMPY
    po_obs = {N.OBS_BOARD: po.OBS_BOARD, N.OBS_FN: po.OBS_FN, N.OBS_RGB: po.OBS_RGB}[obs]
    e = Engine(B, 10, 20, ruleset=ruleset, obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP)
    o = po.OracleBatch(B, 10, 20, ruleset)
    seeds = np.arange(9, 9 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(fuse + B)
    for K in (3, 21, 16):
        acts = rng.integers(0, 8 if numpy_rules else 7, (K, B)).astype(np.int8)
        tobs, trew, tterm, tlines = e.rollout(acts)
        for k in range(K):
            ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            np.testing.assert_array_equal(trew[k], ro, err_msg=f"reward k={k}")
            np.testing.assert_array_equal(tterm[k], to, err_msg=f"terminated k={k}")
            np.testing.assert_array_equal(tlines[k], lo, err_msg=f"lines k={k}")
            np.testing.assert_array_equal(tobs[k], o.observe(po_obs), err_msg=f"obs k={k} (K={K})")
        np.testing.assert_array_equal(e.observation(), o.observe(po_obs))  # the engine's buffer holds the last one
        _cmp_state(e, o, f"K={K}")
    e.close()


@pytest.mark.parametrize("B,T", [(4096, 256), (65536, 192), (524288, 96)])
def test_baseline_batch_sizes_against_oracle(B, T):
    """BASELINE.json's batch sizes (configs 2-4: 4 096, 65 536, 8 x 65 536 envs, 10x20, uniform random actions, env i
    seeded 1 + i, same-step autoreset) on one device, whole trajectories of reward / terminated / lines against the
    oracle plus the final observation and the full state incl. every env's generator."""
    e = Engine(B, 10, 20, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP, episode_stats=True)
    o = po.OracleBatch(B, 10, 20, nthreads=8)
    seeds = np.arange(1, B + 1, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(0)
    n_done = 0
    for t0 in range(0, T, 48):
        K = min(48, T - t0)
        acts = rng.integers(0, 8, (K, B)).astype(np.int8)
        _, rew, term, lines = e.rollout(acts, want_obs=False)
        for k in range(K):
            ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            assert np.array_equal(rew[k], ro) and np.array_equal(term[k], to) and np.array_equal(lines[k], lo), f"t={t0 + k}"
        n_done += int(term.sum())
    np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD))
    ge, go = _cmp_state(e, o, "final")
    for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
        np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f}")
    assert n_done > B // 4  # episodes ended and restarted inside the run (mean episode: 63 transitions)
    e.close()


@pytest.mark.parametrize("autoreset", [N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
@pytest.mark.parametrize("B", [49152, 65536 + 40])
def test_functional_large_batches_one_transition_per_launch(B, autoreset):
    """FUNCTIONAL rule-set on both sides of the batch size (49 152 envs) above which the one-transition kernel requests
    the look-ahead keys behind the publish, by the event lanes only (NEXT_STEP: an env that only resets continues with
    R_n, not R_{n+1}); hard-drop-heavy actions so that locks, bag refills and resets are frequent.  Every transition's
    outputs, the last observation and the full state incl. the jax keys against the oracle."""
    T = 120
    e = Engine(B, 10, 20, ruleset=N.RULESET_FUNCTIONAL, queue_size=7, obs_mode=N.OBS_FN, autoreset=autoreset)
    o = po.OracleBatch(B, 10, 20, po.RULESET_FUNCTIONAL, True, 7, nthreads=8)
    seeds = np.arange(5, 5 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(B)
    n_done = 0
    for t in range(T):
        a = rng.integers(0, 7, B)
        a[rng.random(B) < 0.4] = 6
        got = e.step(a)
        exp = o.step(a.astype(np.int32), autoreset=autoreset)
        for g, x, nm in zip(got, exp, ("reward", "terminated", "lines")):
            assert np.array_equal(g, x), f"{nm} t={t}"
        n_done += int(exp[1].sum())
    np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN))
    ge, go = _cmp_state(e, o, "final")
    np.testing.assert_array_equal(ge["key"], go["key"], err_msg="rng.key")
    assert n_done > B // 2
    e.close()


@pytest.mark.parametrize("lazy", [0, 1, 2, 4, 5])
@pytest.mark.parametrize("autoreset", [N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
@pytest.mark.parametrize("fuse", [1, 16])
def test_functional_reset_bags_sorted_ahead_or_in_line(lazy, autoreset, fuse, monkeypatch):
    """FUNCTIONAL rule-set (envs/tetris_fn.py:318-367, functional/queue.py:20-35): the bag a reset deals is sorted AHEAD only by
    a lock that leaves the stack within reach of the spawn rows (rows 0-5: tg_kernel_bits.h fn_hi); TG_FN_LAZY=1 never sorts
    ahead, so every episode end takes the in-line sort, TG_FN_LAZY=2 always sorts ahead (round 3).  The key-chain look-ahead
    behind a reset is one key deep and catches up over the idle transitions that follow; TG_FN_LAZY bit 2 (4, 5) never
    catches up, so every lock, refill and reset runs its depth-1 form.  All of them against the oracle — outputs of every
    transition, observation, state and jax keys — from fresh resets, from uploaded tall boards (tg_set_state rebuilds the
    whole look-ahead) and from uploaded game-over states (a reset that no lock of the transition precedes), hard-drop-heavy
    so that episodes end all the time."""
    monkeypatch.setenv("TG_FN_LAZY", str(lazy))
    monkeypatch.setenv("TG_FUSE", str(fuse))
    B, W, H = 1500, 10, 20
    e = Engine(B, W, H, ruleset=N.RULESET_FUNCTIONAL, queue_size=7, obs_mode=N.OBS_FN, autoreset=autoreset)
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, True, 7)
    seeds = np.arange(900, 900 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(17 * lazy + fuse + autoreset)
    n_done = 0
    for phase in range(3):
        if phase == 1:  # tall stacks: filled from row 3 / 5 / 7 down except one column — the very next lock can end the game
            b, r, g = o.get_state()
            for i in range(B):
                top = 3 + 2 * (i % 3)
                b[i, top:, :] = 2 + (i % 7)
                b[i, top:, i % W] = 0
            # ... and every eleventh env's piece sits INSIDE the stack: its lock is the literal `board += matrix * id` on filled cells
            r["y"][::11] = 9; r["x"][::11] = 6
            o.set_state(board=b, rec=r, rng=g); e.set_state(board=b, rec=r, rng=g)
        if phase == 2:  # every fifth env is handed over already game over
            b, r, g = o.get_state()
            r["flags"][::5] |= N.F_GAME_OVER
            o.set_state(board=b, rec=r, rng=g); e.set_state(board=b, rec=r, rng=g)
        for K in (1, 7, 16, 40):
            acts = rng.integers(0, 7, (K, B)).astype(np.int8)
            acts[rng.random((K, B)) < 0.45] = 6
            d = e.dev_alloc(acts.nbytes)
            e.dev_upload(d, acts)
            e.step_k_device(d, K)
            for k in range(K):
                ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=autoreset)
                n_done += int(to.sum())
            got = e.fetch_outputs()
            for g_, x, nm in zip(got, (ro, to, lo), ("reward", "terminated", "lines")):
                np.testing.assert_array_equal(g_, x, err_msg=f"{nm} phase={phase} K={K}")
            np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN), err_msg=f"obs phase={phase} K={K}")
            ge, go = _cmp_state(e, o, f"phase={phase} K={K}")
            np.testing.assert_array_equal(ge["key"], go["key"], err_msg=f"rng_key phase={phase} K={K}")
            e.dev_free(d)
    assert n_done > B  # every env ended at least one episode on average
    e.close()


@pytest.mark.parametrize("ruleset", [N.RULESET_NUMPY, N.RULESET_FUNCTIONAL])
@pytest.mark.parametrize("fuse", [1, 8, 16])
def test_fused_rollout_with_frequent_line_clears(ruleset, fuse, monkeypatch):
    """Dense stacks (full rows except a narrow gap) + hard-drop-heavy actions: line clears (mask compaction in the logic
    wave, cooperative uint8 row compaction in the stream wave, the functional rule-set's row-0 quirk) happen all the time
    inside fused groups; outputs of every transition and the final state against the oracle."""
    monkeypatch.setenv("TG_FUSE", str(fuse))
    B, W, H = 768, 10, 20
    numpy_rules = ruleset == N.RULESET_NUMPY
    obs, po_obs = (N.OBS_BOARD, po.OBS_BOARD) if numpy_rules else (N.OBS_FN, po.OBS_FN)
    e = Engine(B, W, H, ruleset=ruleset, obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP)
    o = po.OracleBatch(B, W, H, ruleset)
    seeds = np.arange(100, 100 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(fuse + ruleset)
    hard = 5 if numpy_rules else 6
    total_lines = 0
    for rnd in range(6):
        b = np.zeros((B, H, W), np.uint8)
        for i in range(B):
            top = rng.integers(8, 15)
            b[i, top:] = rng.integers(2, 9, (H - top, W))
            g = rng.integers(0, W - 2)
            b[i, top:, g:g + rng.integers(1, 3)] = 0
            if rng.random() < 0.2:
                b[i, 0, :] = 3 if rng.random() < 0.5 else 0  # occupied top row: the functional clear copies row 0 down
                b[i, 0, rng.integers(0, W)] = 0
        e.set_state(board=b); o.set_state(board=b)
        K = 40
        acts = rng.choice(8 if numpy_rules else 7, (K, B), p=None).astype(np.int8)
        acts[rng.random((K, B)) < 0.35] = hard
        _, rew, term, lines = e.rollout(acts, want_obs=False)
        for k in range(K):
            ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            np.testing.assert_array_equal(lines[k], lo, err_msg=f"lines round {rnd} k={k}")
            np.testing.assert_array_equal(rew[k], ro, err_msg=f"reward round {rnd} k={k}")
            np.testing.assert_array_equal(term[k], to, err_msg=f"terminated round {rnd} k={k}")
        total_lines += int(lines.sum())
        np.testing.assert_array_equal(e.observation(), o.observe(po_obs), err_msg=f"obs round {rnd}")
        _cmp_state(e, o, f"round {rnd}")
    assert total_lines > B // 4  # hundreds of clears inside the fused groups
    e.close()


@pytest.mark.parametrize("W,H,Q,ruleset", [(14, 28, 4, N.RULESET_NUMPY), (10, 22, 4, N.RULESET_NUMPY), (7, 16, 3, N.RULESET_NUMPY),
                                            (13, 20, 7, N.RULESET_NUMPY), (24, 58, 4, N.RULESET_NUMPY),
                                            (7, 14, 4, N.RULESET_NUMPY), (5, 9, 2, N.RULESET_NUMPY), (6, 7, 1, N.RULESET_NUMPY),
                                            (9, 20, 7, N.RULESET_FUNCTIONAL), (14, 28, 7, N.RULESET_FUNCTIONAL),
                                            (7, 14, 7, N.RULESET_FUNCTIONAL)])
def test_board_sizes_compiled_at_create(W, H, Q, ruleset):
    """The reference takes any width / height (envs/tetris.py:77-91, functional/core.py:11-25).  Sizes that are not built
    into the library are compiled for at tg_create (hiprtc, the same kernel templates; odd widths, heights up to 58, slabs
    that are not a multiple of 8, 4 or 2 bytes — 7x14 = 98, 6x7 = 42, 5x9 = 45): single transitions with the reference's dict observation, then fused rollouts,
    against the oracle — rewards, observations, boards, records, generators."""
    numpy_rules = ruleset == N.RULESET_NUMPY
    B, T = 320, 160
    obs = N.OBS_DICT if numpy_rules else N.OBS_FN
    e = Engine(B, W, H, ruleset=ruleset, queue_size=Q, obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP)
    assert e.step_kernel_name().startswith(f"tg::step_kernel_bits<{W}, {H}, {ruleset},")
    o = po.OracleBatch(B, W, H, ruleset, True, Q)
    seeds = np.arange(3, 3 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(W * 100 + H)
    na = 8 if numpy_rules else 7
    n_done = 0
    for t in range(T):
        a = rng.integers(0, na, B)
        if t % 3 == 0:
            a[:] = 5 if numpy_rules else 6  # hard drops: the boards fill up and episodes end
        got = e.step(a)
        exp = o.step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        for g, x, nm in zip(got, exp, ("reward", "terminated", "lines")):
            np.testing.assert_array_equal(g, x, err_msg=f"{nm} t={t}")
        n_done += int(exp[1].sum())
        if t % 8 == 0 or t == T - 1:
            eo, oo = e.observation(), o.observe(po.OBS_DICT if numpy_rules else po.OBS_FN)
            if numpy_rules:
                oo = po.split_dict_obs(oo, B, H + 4, W + 8, 4, Q) if not isinstance(oo, dict) else oo
                for k in eo:
                    np.testing.assert_array_equal(eo[k], oo[k], err_msg=f"obs[{k}] t={t}")
            else:
                np.testing.assert_array_equal(eo, oo, err_msg=f"obs t={t}")
    assert n_done > 0
    for K in (5, 16):
        acts = rng.integers(0, na, (K, B)).astype(np.int8)
        _, rew, term, lines = e.rollout(acts, want_obs=False)
        for k in range(K):
            ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            assert np.array_equal(rew[k], ro) and np.array_equal(term[k], to) and np.array_equal(lines[k], lo), f"K={K} k={k}"
    ge, go = _cmp_state(e, o, "final")
    fields = ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32") if numpy_rules else ("key",)
    for f in fields:
        np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f}")
    e.close()


def test_board_size_that_cannot_be_compiled_is_refused():
    from tetris_gymnasium_amd import EngineError
    with pytest.raises(EngineError, match="outside"):
        Engine(8, 10, 60)


# ---- ActionsMapping / RewardsMapping as per-handle data (ABI 4; reference envs/tetris.py:83-84, 189-190, 461-474) ----------------
class MappedEngineBackend(EngineBackend):
    def __init__(self, batch, width, height, gravity, queue_size, action_ids=None, rewards=None):
        self.e = Engine(batch, width, height, ruleset=N.RULESET_NUMPY, gravity=gravity, queue_size=queue_size,
                        obs_mode=N.OBS_DICT, actions_mapping=action_ids, rewards_mapping=rewards)

    def rewards_f64(self):
        return self.e.host_pack()["rewards"]


@pytest.mark.parametrize("name", ["mapped_10x20_g1", "mapped_10x20_g0", "mapped_clears_10x20", "mapped_alias_left_10x20", "mapped_alias_swap_10x20"])
def test_engine_replays_reference_trace_with_custom_mappings(golden_dir, name):
    """Outputs of the reference's Tetris(actions_mapping=..., rewards_mapping=...): a permutation of the action ids with float
    rewards, a mapping with a duplicate value and ids that no field holds, line clears with integer rewards.  Rewards are
    compared as float32 (tg_step_out) and as the exact Python value (TG_BUF_HOST_PACK)."""
    be = replay.replay_trace(MappedEngineBackend, replay.load(golden_dir, name))
    assert be.e.cfg.flags & (N.FLAG_CUSTOM_ACTIONS | N.FLAG_CUSTOM_REWARDS)


def test_grouped_trace_with_custom_mappings(golden_dir):
    """GroupedActionsObservations over a Tetris with non-stock mappings, terminate_on_illegal_action=False (wrappers/grouped.py:
    225-239): the masked-out action runs the env's own no_op and returns rewards.invalid_action."""
    fx = replay.load(golden_dir, "mapped_grouped_10x20")
    W, H, B, T, Q = int(fx["width"]), int(fx["height"]), int(fx["B"]), int(fx["T"]), int(fx["queue_size"])
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, terminate_on_illegal=bool(fx["terminate_on_illegal"]),
               actions_mapping=fx["action_ids"], rewards_mapping=fx["rewards_mapping"])
    e.reset(1 + np.arange(B, dtype=np.uint64))
    np.testing.assert_array_equal(e.legal_mask(), fx["reset_mask"])
    for t in range(T):
        rew, term, lines = e.step(fx["actions"][t].astype(np.int32))
        np.testing.assert_array_equal(rew, fx["reward_f64"][t].astype(np.float32), err_msg=f"reward t={t}")
        np.testing.assert_array_equal(e.host_pack()["rewards"], fx["reward_f64"][t], err_msg=f"reward (f64) t={t}")
        np.testing.assert_array_equal(term, fx["done"][t], err_msg=f"t={t}")
        np.testing.assert_array_equal(lines, fx["lines"][t])
        np.testing.assert_array_equal(e.get_state()[0], replay.crop(fx["board_after"][t]), err_msg=f"board t={t}")
        if term.any():
            e.reset(mask=term)
        np.testing.assert_array_equal(e.legal_mask(), fx["masks"][t], err_msg=f"mask t={t}")


@pytest.mark.parametrize("autoreset", [N.AUTORESET_NONE, N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
@pytest.mark.parametrize("B,W,H", [(4096, 10, 20), (300, 20, 40)])
def test_custom_mappings_engine_vs_oracle_one_and_sixteen_transitions_per_launch(B, W, H, autoreset):
    """Random permutations of the action ids and random rewards: one transition per launch and the fused rollout (every
    instantiation reads the handle's action table), against the oracle with the same tables."""
    rng = np.random.default_rng(B + autoreset)
    ids = tuple(int(v) for v in rng.permutation(8))
    rws = (float(rng.integers(-3, 4)), 1.0, float(rng.integers(-9, 0)) - 0.5, -0.75)
    e = Engine(B, W, H, obs_mode=N.OBS_BOARD, autoreset=autoreset, actions_mapping=ids, rewards_mapping=rws)
    o = po.OracleBatch(B, W, H, action_ids=ids, rewards=rws)
    seeds = np.arange(5, 5 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    for t in range(60):
        a = rng.integers(0, 8, B)
        re_, te, le = e.step(a)
        ro, to, lo = o.step(a, autoreset=autoreset)
        np.testing.assert_array_equal(re_, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(e.host_pack()["rewards"], o.last_rewards(), err_msg=f"reward f64 t={t}")
        np.testing.assert_array_equal(te, to); np.testing.assert_array_equal(le, lo)
        if autoreset == N.AUTORESET_NONE and te.any():
            e.reset(mask=te); o.reset(mask=te)
    if autoreset != N.AUTORESET_NONE:
        K = 48
        acts = rng.integers(0, 8, (K, B)).astype(np.int8)
        obs, rew, term, lines = e.rollout(acts)
        for k in range(K):
            ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=autoreset)
            np.testing.assert_array_equal(rew[k], ro, err_msg=f"rollout reward k={k}")
            np.testing.assert_array_equal(term[k], to); np.testing.assert_array_equal(lines[k], lo)
            np.testing.assert_array_equal(obs[k], o.observe(po.OBS_BOARD), err_msg=f"rollout obs k={k}")
    _cmp_state(e, o, "end")


def test_tetris_class_with_custom_mappings_returns_the_references_python_rewards(golden_dir):
    from tetris_gymnasium_amd.envs.tetris import Tetris
    from tetris_gymnasium_amd.mappings.actions import ActionsMapping
    from tetris_gymnasium_amd.mappings.rewards import RewardsMapping

    fx = replay.load(golden_dir, "mapped_10x20_g1")
    env = Tetris(actions_mapping=ActionsMapping(*[int(v) for v in fx["action_ids"]]),
                 rewards_mapping=RewardsMapping(*[float(v) for v in fx["rewards_mapping"]]))
    assert env.actions.hard_drop == 2 and env.reward_range == (-5.25, 10.0)
    for i in (0, 7):
        obs, _ = env.reset(seed=1 + i)
        assert replay.obs_crc({k: v[None] for k, v in obs.items()}, 0) == int(fx["first_obs_crc"][i])
        for t in range(200):
            obs, r, d, _, info = env.step(int(fx["actions"][t, i]))
            assert r == fx["reward_f64"][t, i] and d == bool(fx["done"][t, i]) and info["lines_cleared"] == fx["lines"][t, i]
            assert replay.obs_crc({k: v[None] for k, v in obs.items()}, 0) == int(fx["crc"][t, i])
            if d:
                env.reset()
    env.close()


# ---- EnvConfig.queue_size as the bag length of the functional rule-set (functional/queue.py:35, 67) ---------------------------------
@pytest.mark.parametrize("autoreset", [N.AUTORESET_NONE, N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
@pytest.mark.parametrize("k,B", [(1, 200), (2, 1000), (3, 4096), (5, 300), (6, 1000)])
def test_functional_bag_length_engine_vs_oracle(k, B, autoreset, monkeypatch):
    """queue_size k < 7: bags are permutations of 0..k-1 (sorts of k elements in the look-ahead pass, refill at queue_index >= k);
    one transition per launch and fused rollouts, the in-line sort paths included (TG_FN_LAZY=1 for one size)."""
    if k == 3:
        monkeypatch.setenv("TG_FN_LAZY", "1")
    e = Engine(B, 10, 20, ruleset=N.RULESET_FUNCTIONAL, queue_size=k, obs_mode=N.OBS_FN, autoreset=autoreset)
    o = po.OracleBatch(B, 10, 20, po.RULESET_FUNCTIONAL, queue_size=k)
    seeds = np.arange(3, 3 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN))
    rng = np.random.default_rng(k * 10 + autoreset)
    for t in range(80):
        a = rng.integers(0, 7, B)
        re_, te, le = e.step(a)
        ro, to, lo = o.step(a, autoreset=autoreset)
        np.testing.assert_array_equal(re_, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(te, to, err_msg=f"terminated t={t}"); np.testing.assert_array_equal(le, lo)
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN), err_msg=f"obs t={t}")
        if autoreset == N.AUTORESET_NONE and te.any():
            e.reset(mask=te); o.reset(mask=te)
    if autoreset != N.AUTORESET_NONE:
        K = 64
        acts = rng.integers(0, 7, (K, B)).astype(np.int8)
        obs, rew, term, lines = e.rollout(acts)
        for j in range(K):
            ro, to, lo = o.step(acts[j].astype(np.int32), autoreset=autoreset)
            np.testing.assert_array_equal(rew[j], ro, err_msg=f"rollout reward j={j}")
            np.testing.assert_array_equal(term[j], to); np.testing.assert_array_equal(obs[j], o.observe(po.OBS_FN), err_msg=f"rollout obs j={j}")
    ge, go = _cmp_state(e, o, "end")
    np.testing.assert_array_equal(ge["key"], go["key"])
    _, rec, _ = e.get_state()
    assert (rec["piece"] < k).all() and (rec["bag"][:, k:] == 0).all()
    # a state round trip rebuilds the look-ahead for this bag length (bits_prepare_kernel)
    e.set_state(*e.get_state())
    a = rng.integers(0, 7, B)
    np.testing.assert_array_equal(e.step(a)[0], o.step(a, autoreset=autoreset)[0])
    e.close()


@pytest.mark.parametrize("gravity", [False, True])
@pytest.mark.parametrize("fuse", [1, 16])
def test_functional_piece_uploaded_on_top_of_filled_cells(gravity, fuse):
    """A state only an upload makes (found by tests/fuzz_parity.py): the active piece of the functional rule-set lying ON filled
    cells.  It stays there while nothing moves it (every move into a collision is refused: envs/tetris_fn.py:161-273; with gravity
    off for as long as the caller likes), the observation is (board > 0) + (-1) * piece cell by cell — 1 - 1 = 0 under the piece
    (envs/tetris_fn.py:148-155) — and un-drawing the piece inside a launch of several transitions must give the covered cells back.
    Also: TG_BUF_OBS shows the uploaded state right after tg_set_state, for the BOARD / FN surfaces too."""
    B, W, H = 96, 10, 20
    e = Engine(B, W, H, ruleset=N.RULESET_FUNCTIONAL, gravity=gravity, queue_size=7, obs_mode=N.OBS_FN, fuse=fuse)
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, gravity, 7)
    seeds = np.arange(21, 21 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(5)
    for round_ in range(3):
        b, _, _ = o.get_state()
        b[:] = np.where(rng.random(b.shape) < 0.5, rng.integers(2, 9, b.shape), 0)
        b[:, :4] = np.where(rng.random((B, 4, W)) < 0.7, 3, 0)  # dense where the pieces spawn: most of them lie on filled cells
        b[B // 2:, :3] = 0                                       # ... the other half of the batch plays on
        e.set_state(board=b); o.set_state(board=b)
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_FN), err_msg="TG_BUF_OBS after tg_set_state")
        acts = rng.integers(0, 7, (24, B))
        obs, rew, term, lines = e.rollout(acts.astype(np.int8))
        for k in range(24):
            ro, to, lo = o.step(acts[k])
            np.testing.assert_array_equal(obs[k], o.observe(po.OBS_FN), err_msg=f"observation of transition {k}, round {round_}")
            np.testing.assert_array_equal(term[k], to); np.testing.assert_array_equal(rew[k], ro)
        be, re_, _ = e.get_state(); bo, ro_, _ = o.get_state()
        np.testing.assert_array_equal(be, bo)
        for f in ("piece", "rot", "x", "y", "flags", "score"):
            np.testing.assert_array_equal(re_[f], ro_[f], err_msg=f)
    e.reset(seeds); o.reset(seeds)  # (a full reset: the plain draw is back — nothing to compare but the results)
    acts = rng.integers(0, 7, (16, B))
    obs, rew, term, lines = e.rollout(acts.astype(np.int8))
    for k in range(16):
        o.step(acts[k])
        np.testing.assert_array_equal(obs[k], o.observe(po.OBS_FN))
    e.close()
