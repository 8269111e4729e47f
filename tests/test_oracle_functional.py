"""Functional rule-set of the oracle (functional/core.py, envs/tetris_fn.py restated): the reference's own
known-answer tests (tests/test_functional/**) restated against it, plus a cross-check of the mechanics it
shares with the NumPy rule-set (collision, drop distance, lock-on-blocked-gravity, compaction) through the
SURVEY §8.0 action mapping.  jax is not installable here, so this is how the functional logic is pinned."""
import numpy as np
import pytest

from oracle import pyoracle as po

W, H, P = 10, 20, 4


def make(gravity=True, B=1, key=42):
    o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, gravity)
    o.reset(seeds=np.full(B, key, np.uint64))  # PRNGKey(42) = [0, 42]
    return o


def put(o, board=None, **fields):
    b, rec, rng = o.get_state()
    for k, v in fields.items():
        rec[k][:] = v
    if "piece" in fields and "val" not in fields:
        rec["val"][:] = np.asarray(fields["piece"]) + 2
    o.set_state(board=b if board is None else board, rec=rec)


def test_scoring_table():  # tests/test_functional/test_core/test_scoring.py:11-17
    assert [po.lib().tgo_fn_score(n) for n in range(5)] == [0, 100, 300, 500, 800]


def test_lut_shape_and_ids():  # tests/test_functional/test_tetrominoes.py:13-19
    import ctypes as C
    for p in range(7):
        cells = []
        for r in range(4):
            m = np.zeros(16, np.uint8)
            po.lib().tgo_fn_matrix(p, r, m.ctypes.data_as(C.c_void_p))
            assert m.sum() == 4 and set(np.unique(m)) <= {0, 1}
            cells.append(m.reshape(4, 4))
        # rotation r+1 = np.rot90(k=1) of rotation r within the native box
        n = {0: 4, 1: 2}.get(p, 3)
        for r in range(3):
            np.testing.assert_array_equal(cells[r + 1][:n, :n], np.rot90(cells[r][:n, :n]))


def test_reset_state():  # test_env/test_reset.py + test_core/test_board.py:7-44
    o = make()
    board, rec, rng = o.get_state()
    assert board.shape == (1, 20, 10) and not board.any()
    assert o.padded_boards().shape == (1, 24, 18)
    pb = o.padded_boards()[0]
    assert (pb[20:] == 1).all() and (pb[:, :4] == 1).all() and (pb[:, 14:] == 1).all()
    assert rec["x"][0] == (W + 2 * P) // 2 - 2 == 7 and rec["y"][0] == 0 and rec["rot"][0] == 0
    assert sorted(rec["bag"][0]) == list(range(7)) and rec["bag_index"][0] == 1
    assert rec["piece"][0] == rec["bag"][0][0]
    assert rec["score"][0] == 0.0 and not rec["flags"][0] & 1
    # same key → same state (test_reset.py:30-45)
    o2 = make()
    assert o2.get_state()[1].tobytes() == rec.tobytes()


@pytest.mark.parametrize("rows,expect", [([19], 1), ([19, 18], 2), ([19, 18, 17, 16], 4)])
def test_line_clear_counts(rows, expect):  # test_core/test_line_clear.py:14-60
    o = make(gravity=False)
    b = np.zeros((1, H, W), np.uint8)
    for r in rows:
        b[0, r, :] = 2
    # lock an O piece far away from the filled rows (top-left), hard drop with no gravity
    put(o, board=b, piece=1, rot=0, x=4 + 4, y=0)
    rew, term, lines = o.step(np.array([6]))
    assert lines[0] == expect
    nb = o.get_state()[0][0]
    assert (nb > 0).sum() == 4  # only the O piece remains
    dist = H - len(rows) - 2  # O (rows y..y+1) lands on top of the filled rows before compaction
    assert rew[0] == 2 * dist + [0, 100, 300, 500, 800][expect]


def test_partial_row_not_cleared_and_marker_shifts():  # test_line_clear.py:62-71 + :73-90
    o = make(gravity=False)
    b = np.zeros((1, H, W), np.uint8)
    b[0, 19, :9] = 2
    put(o, board=b, piece=1, rot=0, x=4, y=0)
    _, _, lines = o.step(np.array([6]))
    assert lines[0] == 0
    o = make(gravity=False)
    b = np.zeros((1, H, W), np.uint8)
    b[0, 17, 0] = 5  # marker
    b[0, 19, :] = 2
    b[0, 19, 8:10] = 0
    b[0, 18, 8:10] = 0
    put(o, board=b, piece=1, rot=0, x=4 + 8, y=0)  # O fills cols 8,9 rows 18,19 → row 19 full
    _, _, lines = o.step(np.array([6]))
    assert lines[0] == 1
    nb = o.get_state()[0][0]
    assert nb[18, 0] == 5 and nb[17, 0] == 0
    assert nb[19, 8] == 3 and nb[19, 9] == 3  # the half of the O from row 18 moved down


def test_gravity_blocked_by_piece():  # test_core/test_gravity.py:28-43 (T at y=10 blocks T at y=8)
    o = make(gravity=True)
    b = np.zeros((1, H, W), np.uint8)
    for (r, c) in [(10, 4), (11, 3), (11, 4), (11, 5)]:  # T matrix at (x=7,y=10) → playfield col x-4+j
        b[0, r, c] = 4
    put(o, board=b, piece=2, rot=0, x=7, y=7)
    o.step(np.array([5]))  # no-op: gravity only
    _, rec, _ = o.get_state()
    assert rec["y"][0] == 8
    o.step(np.array([5]))  # at y=8 gravity is blocked → lock + spawn
    b2, rec, _ = o.get_state()
    assert rec["y"][0] == 0 and (b2[0] > 0).sum() == 8


def test_hard_drop_reward_twice_distance():  # test_core/test_hard_drop.py:34-39
    o = make(gravity=False)
    put(o, piece=0, rot=0, x=7, y=0)
    rew, _, _ = o.step(np.array([6]))
    assert rew[0] == 2 * 18  # I row 1 of the matrix lands on row 19 → y = 18


def test_rotation_actions():  # test_env/test_rotation.py:9-19 (4 = rot+1, 3 = rot-1)
    o = make(gravity=False)
    put(o, piece=2, rot=0, x=7, y=5)
    o.step(np.array([4])); assert o.get_state()[1]["rot"][0] == 1
    o.step(np.array([3])); assert o.get_state()[1]["rot"][0] == 0
    o.step(np.array([3])); assert o.get_state()[1]["rot"][0] == 3


def test_movement_and_soft_drop():  # test_env/test_movement.py
    o = make(gravity=False)
    put(o, piece=2, rot=0, x=7, y=5)
    o.step(np.array([0])); assert o.get_state()[1]["x"][0] == 6
    o.step(np.array([1])); o.step(np.array([1])); assert o.get_state()[1]["x"][0] == 8
    rew, _, _ = o.step(np.array([2]))
    assert o.get_state()[1]["y"][0] == 6 and rew[0] == 1.0
    o = make(gravity=True)
    put(o, piece=2, rot=0, x=7, y=5)
    rew, _, _ = o.step(np.array([2]))
    assert o.get_state()[1]["y"][0] == 7 and rew[0] == 1.0  # move down + gravity


def test_game_over_is_identity():  # test_env/test_step.py:16-25
    o = make()
    b, rec, rng = o.get_state()
    rec["flags"] |= 1
    o.set_state(rec=rec)
    before = o.get_state()
    rew, term, lines = o.step(np.array([0]))
    after = o.get_state()
    assert term[0] == 1 and rew[0] == 0.0 and lines[0] == 0
    assert all(a.tobytes() == b_.tobytes() for a, b_ in zip(before, after))


def test_observation_values():  # test_env/test_observations.py:13-60
    o = make()
    obs = o.observe(po.OBS_FN)
    assert obs.shape == (1, 20, 10) and obs.dtype == np.int8
    assert set(np.unique(obs)) <= {-1, 0, 1} and (obs == -1).sum() == 4
    rng = np.random.default_rng(0)
    for _ in range(300):
        _, term, _ = o.step(rng.integers(0, 7, 1))
        obs = o.observe(po.OBS_FN)
        assert set(np.unique(obs)) <= {-1, 0, 1}
        if term[0]:
            assert (obs == -1).sum() == 0  # no active piece shown when game over
            break
    else:
        pytest.fail("random play did not reach game over")


def test_integration_score_monotone_and_deterministic():  # test_integration.py:20-69
    acts = np.random.default_rng(5).integers(0, 7, 2000)
    runs = []
    for _ in range(2):
        o = make()
        last, over_at = 0.0, None
        for t, a in enumerate(acts):
            _, term, _ = o.step(np.array([a]))
            s = float(o.get_state()[1]["score"][0])
            assert s >= last
            last = s
            if term[0]:
                over_at = t
                break
        runs.append((over_at, o.get_state()[0].tobytes(), last))
    assert runs[0] == runs[1] and runs[0][0] is not None


def test_batched_equals_sequential():  # test_batched.py:64-89
    B = 4
    keys = np.array([1, 2, 3, 4], np.uint64)
    ob = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, True)
    ob.reset(seeds=keys)
    singles = []
    for k in keys:
        o = po.OracleBatch(1, W, H, po.RULESET_FUNCTIONAL, True)
        o.reset(seeds=np.array([k], np.uint64))
        singles.append(o)
    acts = np.random.default_rng(1).integers(0, 7, (200, B))
    for a in acts:
        rb = ob.step(a)
        for i, o in enumerate(singles):
            ri = o.step(a[i:i + 1])
            assert all(x[i] == y[0] for x, y in zip(rb, ri))
    sb = ob.get_state()
    for i, o in enumerate(singles):
        si = o.get_state()
        assert sb[0][i].tobytes() == si[0][0].tobytes() and sb[1][i].tobytes() == si[1][0].tobytes()


NP2FN = {0: 0, 1: 1, 2: 2, 3: 4, 4: 3, 5: 6, 7: 5}


@pytest.mark.parametrize("gravity", [True, False])
def test_shared_mechanics_match_numpy_ruleset(gravity):
    """Same board/piece/position, mapped action: both rule-sets must agree on the resulting board, lines,
    and (when no lock happened) x/y/rot — except where SURVEY §8.0 says they differ by design."""
    B = 64
    on = po.OracleBatch(B, W, H, po.RULESET_NUMPY, gravity)
    of = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL, gravity)
    on.reset(seeds=1 + np.arange(B, dtype=np.uint64))
    of.reset(seeds=1 + np.arange(B, dtype=np.uint64))
    rng = np.random.default_rng(3)
    # start from half-filled boards so locks, overhang drops and line clears all occur
    b0 = (rng.random((B, H, W)) < 0.45).astype(np.uint8) * rng.integers(2, 9, (B, H, W)).astype(np.uint8)
    b0[:, :8] = 0
    for i in range(B // 2, B):  # second half: dense stacks with a 2-3 wide gap under the spawn column
        b0[i, 12:] = rng.integers(2, 9, (H - 12, W))
        g = int(rng.integers(2, 5))
        b0[i, :, g:g + int(rng.integers(2, 4))] = 0
    on.set_state(board=b0)
    locks = clears = 0
    for t in range(600):
        bn, rn, _ = on.get_state()
        alive = (rn["flags"] & 1) == 0
        _, rf, gf = of.get_state()
        for k in ("piece", "rot", "x", "y", "val"):
            rf[k] = rn[k]
        rf["flags"] = 0
        of.set_state(board=bn, rec=rf)
        a_np = rng.choice([0, 1, 2, 3, 4, 5, 7], B)
        a_fn = np.array([NP2FN[a] for a in a_np])
        rwn, tn, ln = on.step(a_np)
        rwf, tf, lf = of.step(a_fn)
        bn2, rn2, _ = on.get_state()
        bf2, rf2, _ = of.get_state()
        locked = rwn > 0
        # NumPy "commit while already colliding" (envs/tetris.py:461-463) has no functional counterpart
        ok = alive & ~((tn == 1) & (rwn == 0) & (ln == 0) & (bn2.reshape(B, -1) == bn.reshape(B, -1)).all(1))
        np.testing.assert_array_equal(bn2[ok], bf2[ok])
        np.testing.assert_array_equal(ln[ok], lf[ok])
        moved = ok & ~locked & (tn == 0)
        for k in ("x", "y", "rot", "piece"):
            np.testing.assert_array_equal(rn2[k][moved], rf2[k][moved], err_msg=k)
        locks += int(locked.sum()); clears += int(ln.sum())
        on.reset(mask=tn)
        if t % 20 == 19:
            bb, _, _ = on.get_state()
            bb[:] = b0
            on.set_state(board=bb)
    assert locks > 500 and clears > 20


@pytest.mark.parametrize("k", [1, 2, 3, 5, 7])
def test_bag_length_is_config_queue_size(k):
    """functional/queue.py:35, 67: the bag is permutation(arange(queue_size)) and is refilled at queue_index >= queue_size — with
    queue_size k < 7 only the first k tetrominoes are ever dealt, every k consecutive spawns are a permutation of 0..k-1."""
    B = 16
    o = po.OracleBatch(B, 10, 20, po.RULESET_FUNCTIONAL, queue_size=k)
    o.reset(np.arange(1, B + 1, dtype=np.uint64))
    _, rec, _ = o.get_state()
    assert (rec["bag_index"] == 1).all() and (rec["piece"] == rec["bag"][:, 0]).all() and (rec["bag"][:, k:] == 0).all()
    for b in rec["bag"]:
        assert sorted(b[:k]) == list(range(k))
    spawned = [rec["piece"].copy()]
    for t in range(6 * k):
        o.step(np.full(B, 6, np.int32), autoreset=po.AUTORESET_SAME_STEP)  # hard drop: a lock and a spawn per step
        _, rec, _ = o.get_state()
        assert (rec["piece"] < k).all() and (rec["bag_index"] >= 1).all() and (rec["bag_index"] <= k).all()
        spawned.append(rec["piece"].copy())
    seq = np.stack(spawned, 1)  # [B][1 + 6k]; the first k spawns of an episode are one whole bag
    for i in range(B):
        assert sorted(seq[i, :k]) == list(range(k))
