"""GPU parity of the grouped action set (wrappers/grouped.py: column*4+rotation → teleport + hard drop) against
reference-generated fixtures and the oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay

pytestmark = pytest.mark.gpu

from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402


@pytest.mark.parametrize("name", ["grouped_10x20", "grouped_20x40", "grouped_clears_10x20"])
def test_grouped_trace_against_reference_fixture(golden_dir, name):
    fx = replay.load(golden_dir, name)
    W, H, B, T, Q = int(fx["width"]), int(fx["height"]), int(fx["B"]), int(fx["T"]), int(fx["queue_size"])
    init = fx["init_boards"] if "init_boards" in fx.files and fx["init_boards"].size else None
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True)
    e.reset(1 + np.arange(B, dtype=np.uint64))
    if init is not None:
        e.set_state(board=replay.crop(init))
    np.testing.assert_array_equal(e.legal_mask(), fx["reset_mask"])
    for t in range(T):
        rew, term, lines = e.step(fx["actions"][t].astype(np.int32))
        np.testing.assert_array_equal(rew, fx["reward"][t], err_msg=f"reward t={t}")  # -0.1f for illegal actions
        np.testing.assert_array_equal(term, fx["done"][t], err_msg=f"t={t}")
        np.testing.assert_array_equal(lines, fx["lines"][t])
        np.testing.assert_array_equal(e.get_state()[0], replay.crop(fx["board_after"][t]), err_msg=f"board t={t}")
        if term.any():
            e.reset(mask=term)
            if init is not None:
                b = e.get_state()[0]
                b[term.astype(bool)] = replay.crop(init)[term.astype(bool)]
                e.set_state(board=b)
        np.testing.assert_array_equal(e.legal_mask(), fx["masks"][t], err_msg=f"mask t={t}")


@pytest.mark.parametrize("term_on_illegal", [True, False])
@pytest.mark.parametrize("W,H,Q", [(10, 20, 4), (20, 40, 7)])
@pytest.mark.parametrize("autoreset", [N.AUTORESET_NONE, N.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP])
def test_grouped_engine_vs_oracle(W, H, Q, term_on_illegal, autoreset):
    B, T = 256, 150
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, terminate_on_illegal=term_on_illegal,
               autoreset=autoreset)
    o = po.OracleBatch(B, W, H, queue_size=Q)
    seeds = np.arange(7, 7 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(W + autoreset)
    n_ill = 0
    for t in range(T):
        np.testing.assert_array_equal(e.legal_mask(), o.grouped_enumerate(), err_msg=f"mask t={t}")
        a = rng.integers(0, 4 * W, B)  # uniformly over ALL actions: ~10 % illegal
        re_, te, le = e.step(a)
        ro, to, lo, ill = o.grouped_step(a, terminate_on_illegal=term_on_illegal, autoreset=autoreset)
        np.testing.assert_array_equal(re_, ro, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(te, to, err_msg=f"terminated t={t}")
        np.testing.assert_array_equal(le, lo)
        np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD), err_msg=f"obs t={t}")
        n_ill += int(ill.sum())
        if autoreset == N.AUTORESET_NONE and te.any():
            e.reset(mask=te); o.reset(mask=te)
    be, re2, _ = e.get_state(); bo, ro2, _ = o.get_state()
    np.testing.assert_array_equal(be, bo)
    for k in re2.dtype.names:
        if k != "reserved":
            np.testing.assert_array_equal(re2[k], ro2[k], err_msg=k)
    assert n_ill > 0


def test_device_side_legal_sampling():
    B, W = 4096, 20
    e = Engine(B, W, 40, queue_size=7, obs_mode=N.OBS_BOARD, grouped=True, autoreset=N.AUTORESET_SAME_STEP)
    e.reset(np.arange(1, B + 1, dtype=np.uint64))
    seen = np.zeros(4 * W, int)
    for t in range(60):
        mask = e.legal_mask()
        a = e.sample_legal_actions(seed=123, counter=t)
        assert mask[np.arange(B), a].all(), "sampled an illegal action"
        seen += np.bincount(a, minlength=4 * W)
        a2 = e.sample_legal_actions(seed=123, counter=t)
        np.testing.assert_array_equal(a, a2)  # deterministic in (seed, counter, env)
        e.step_sampled()
        rew, term, lines = e.fetch_outputs()
        assert (rew >= 0).all()  # never the -0.1 penalty
    assert (seen > 0).sum() > 3 * W  # covers most of the action space


@pytest.mark.parametrize("name", ["grouped_features_10x20", "grouped_features_20x40", "grouped_features_prefull_6x12",
                                  "grouped_features_prefull_10x20"])
def test_placement_features_against_reference_fixture(golden_dir, name):
    """[4W][W+3] feature tensor of GroupedActionsObservations + FeatureVectorObservation, computed on bit masks."""
    fx = replay.load(golden_dir, name)
    W, H, Q, n = int(fx["width"]), int(fx["height"]), int(fx["queue_size"]), fx["boards"].shape[0]
    e = Engine(n, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, features=True)
    e.reset(1 + np.arange(n, dtype=np.uint64))
    _, rec, _ = e.get_state()
    for i in range(n):
        val = int(fx["mats"][i].max())
        rec["rot"][i] = next(r for r in range(4) if np.array_equal(replay.matrix16(fx["pieces"][i], r, val), fx["mats"][i]))
        rec["piece"][i], rec["val"][i] = fx["pieces"][i], val
    rec["x"] = (W + 8) // 2 - np.where(fx["pieces"] == 0, 2, 1)
    rec["y"] = 0
    e.set_state(board=replay.crop(fx["boards"]), rec=rec)
    pf, cf = e.features()
    np.testing.assert_array_equal(pf, fx["feats"])
    np.testing.assert_array_equal(cf, fx["cur_feats"])
    np.testing.assert_array_equal(e.legal_mask(), fx["masks"])
    if "pboards" in fx:  # (the *_prefull fixtures: rows that were full BEFORE the placement leave with every placement, envs/tetris.py:493)
        np.testing.assert_array_equal(e.placement_boards()[1], fx["pboards"])


@pytest.mark.parametrize("W,H", [(10, 20), (7, 14), (6, 9)])
def test_placement_features_engine_vs_oracle_random_states(W, H):
    B = 512
    e = Engine(B, W, H, obs_mode=N.OBS_BOARD, grouped=True, features=True, autoreset=N.AUTORESET_SAME_STEP)
    o = po.OracleBatch(B, W, H)
    seeds = np.arange(3, 3 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(0)
    for t in range(40):
        if t % 8 == 0:  # dense stacks: placements that clear 1-4 rows
            b = rng.integers(2, 9, (B, H, W)).astype(np.uint8)
            b[:, : rng.integers(2 * H // 5, 4 * H // 5)] = 0
            for i in range(B):
                g = rng.integers(0, W - 3); b[i, :, g:g + rng.integers(1, 4)] = 0
            e.set_state(board=b); o.set_state(board=b)
        pf, cf = e.features()
        np.testing.assert_array_equal(pf, o.grouped_features(), err_msg=f"placement features t={t}")
        np.testing.assert_array_equal(cf, o.features(), err_msg=f"current features t={t}")
        a = e.sample_legal_actions(seed=5, counter=t)
        e.step_sampled(); o.grouped_step(a, autoreset=po.AUTORESET_SAME_STEP)


def _poke_piece(e, pieces, mats):
    _, rec, _ = e.get_state()
    for i in range(len(pieces)):
        val = int(mats[i].max())
        rec["rot"][i] = next(r for r in range(4) if np.array_equal(replay.matrix16(pieces[i], r, val), mats[i]))
        rec["piece"][i], rec["val"][i] = pieces[i], val
    return rec


@pytest.mark.parametrize("name", ["grouped_10x20", "grouped_20x40", "grouped_clears_10x20"])
def test_placement_boards_against_reference_fixture(golden_dir, name):
    """The wrapper's observation tensor [4W][Hp][Wp] (wrappers/grouped.py:124-207) built on the device, against the
    boards the reference produced for recorded states (illegal → ones, game-over placement → zeros, row clears)."""
    fx = replay.load(golden_dir, name)
    n = fx["enum_mask"].shape[0]
    W, H = int(fx["width"]), int(fx["height"])
    e = Engine(n, W, H, queue_size=int(fx["queue_size"]), obs_mode=N.OBS_BOARD, grouped=True)
    e.reset(1 + np.arange(n, dtype=np.uint64))
    e.set_state(board=replay.crop(fx["enum_state_board"]), rec=_poke_piece(e, fx["enum_piece"], fx["enum_matrix"]))
    mask, boards = e.placement_boards()
    np.testing.assert_array_equal(mask, fx["enum_mask"])
    np.testing.assert_array_equal(boards, fx["enum_boards"])


def test_placement_boards_reference_csv(golden_dir):
    """The reference's only golden file: tests/test_grouped_env/expected_result_i_placement.csv = board of action 5*4+1
    on the mock board with a vertical I (values 1), and its 40 placement boards."""
    fx = replay.load(golden_dir, "ref_tests_grouped")
    e = Engine(1, 10, 20, obs_mode=N.OBS_BOARD, grouped=True)
    e.reset(np.array([42], np.uint64))
    vi = np.zeros((4, 4), np.uint8); vi[:4, :4] = np.rot90(replay.NATIVE[0])
    e.set_state(board=replay.crop(fx["mock_board"][None]), rec=_poke_piece(e, np.array([0]), vi.reshape(1, 16)))
    mask, boards = e.placement_boards()
    np.testing.assert_array_equal(mask[0], fx["mask"])
    np.testing.assert_array_equal(boards[0], fx["boards40"])
    np.testing.assert_array_equal(boards[0, 5 * 4 + 1], fx["csv_board"])


@pytest.mark.parametrize("W,H,Q", [(10, 20, 4), (20, 40, 7), (10, 24, 4), (6, 12, 2), (7, 14, 4), (5, 9, 2)])
def test_placement_boards_engine_vs_oracle_random_states(W, H, Q):
    B = 384
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, autoreset=N.AUTORESET_SAME_STEP)
    o = po.OracleBatch(B, W, H, queue_size=Q)
    seeds = np.arange(11, 11 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    rng = np.random.default_rng(W)
    for t in range(24):
        if t % 6 == 0:  # dense stacks: placements that clear 1-4 rows, and stacks reaching the spawn rows
            b = rng.integers(2, 9, (B, H, W)).astype(np.uint8)
            b[:, : rng.integers(1, H - 4)] = 0
            for i in range(B):
                g = rng.integers(0, W - 3); b[i, :, g:g + rng.integers(1, 4)] = 0
            e.set_state(board=b); o.set_state(board=b)
        mask, boards = e.placement_boards()
        om, ob = o.grouped_enumerate(boards=True)
        np.testing.assert_array_equal(mask, om, err_msg=f"mask t={t}")
        np.testing.assert_array_equal(boards, ob, err_msg=f"boards t={t}")
        a = e.sample_legal_actions(seed=9, counter=t)
        e.step_sampled(); o.grouped_step(a, autoreset=po.AUTORESET_SAME_STEP)


@pytest.mark.parametrize("kernel_fuse", [1, 8])
@pytest.mark.parametrize("W,H,Q,B", [(20, 40, 7, 1000), (10, 20, 4, 2500)])
def test_grouped_rollout_equals_sample_then_step(W, H, Q, B, kernel_fuse, monkeypatch):
    """tg_grouped_rollout draws the random legal placement inside the step kernel (up to 8 transitions per launch):
    actions, outputs and final state must equal K x {tg_grouped_sample; tg_step} on a second engine, and the oracle
    stepped with the drawn actions."""
    monkeypatch.setenv("TG_FUSE", str(kernel_fuse))
    kw = dict(queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, autoreset=N.AUTORESET_SAME_STEP)
    a_eng, b_eng = Engine(B, W, H, **kw), Engine(B, W, H, **kw)
    o = po.OracleBatch(B, W, H, queue_size=Q)
    seeds = np.arange(31, 31 + B, dtype=np.uint64)
    a_eng.reset(seeds); b_eng.reset(seeds); o.reset(seeds)
    counter = 0
    for K in (1, 5, 19, 32):
        acts, rew, term, lines = a_eng.grouped_rollout(K, seed=77, counter0=counter)
        for k in range(K):
            mask = b_eng.legal_mask()
            a = b_eng.sample_legal_actions(seed=77, counter=counter + k)
            np.testing.assert_array_equal(acts[k], a, err_msg=f"drawn action k={k} (K={K})")
            assert mask[np.arange(B), a].all()
            b_eng.step_sampled()
            rb, tb, lb = b_eng.fetch_outputs()
            ro, to, lo, ill = o.grouped_step(a.astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            for got, ref, ora, nm in ((rew[k], rb, ro, "reward"), (term[k], tb, to, "terminated"), (lines[k], lb, lo, "lines")):
                np.testing.assert_array_equal(got, ref, err_msg=f"{nm} k={k}")
                np.testing.assert_array_equal(got, ora, err_msg=f"{nm} vs oracle k={k}")
            assert not ill.any()
        counter += K
        np.testing.assert_array_equal(a_eng.observation(), o.observe(po.OBS_BOARD))
        ba, ra, ga = a_eng.get_state(); bo, ro2, go = o.get_state()
        np.testing.assert_array_equal(ba, bo)
        for f in ra.dtype.names:
            if f != "reserved":
                np.testing.assert_array_equal(ra[f], ro2[f], err_msg=f)
        np.testing.assert_array_equal(ga["state_lo"], go["state_lo"])
    a_eng.close(); b_eng.close()


def test_config5_full_size_against_oracle():
    """BASELINE.json config 5 at its full size: 65 536 envs, 20x40 board, queue_size 7, grouped hard-drop action set
    (wrappers/grouped.py:209-269), uniform random legal placements drawn inside the step kernel (tg_grouped_rollout,
    8 transitions per launch), env i seeded 1 + i, same-step autoreset.  Whole trajectory of drawn actions (legal under
    the oracle's mask) / reward / terminated / lines against OracleBatch.grouped_step, then the final observation, boards,
    records and every env's generator."""
    B, W, H, Q, T = 65536, 20, 40, 7, 40
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True, autoreset=N.AUTORESET_SAME_STEP)
    o = po.OracleBatch(B, W, H, queue_size=Q, nthreads=8)
    seeds = np.arange(1, B + 1, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    _tall_boards(e, o, every=4)  # 40 placements never fill an empty 20x40 board: a quarter of the envs start near the top
    n_done = n_lock = 0
    for t0 in range(0, T, 20):
        K = min(20, T - t0)
        acts, rew, term, lines = e.grouped_rollout(K, seed=5, counter0=t0)
        for k in range(K):
            if k % 10 == 0:
                om = o.grouped_enumerate()
                assert om[np.arange(B), acts[k]].all(), f"illegal draw at t={t0 + k}"
            ro, to, lo, ill = o.grouped_step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            assert not ill.any()
            assert np.array_equal(rew[k], ro) and np.array_equal(term[k], to) and np.array_equal(lines[k], lo), f"t={t0 + k}"
            n_lock += int((ro > 0).sum())
        n_done += int(term.sum())
    np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD))
    be, re_, ge = e.get_state(); bo, ro2, go = o.get_state()
    np.testing.assert_array_equal(be, bo)
    for f in re_.dtype.names:
        if f != "reserved":
            np.testing.assert_array_equal(re_[f], ro2[f], err_msg=f)
    for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
        np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f}")
    assert n_lock > B * T // 2  # every grouped step is a hard drop + lock
    assert n_done > B // 8      # ... and episodes ended: same-step resets (reset shuffle + refilled queue) ran at this grid size
    e.close()


def _tall_boards(e, o, every=1, top=9, seed=3):
    """Give every `every`-th env (engine and oracle alike) a board whose rows top.. are filled except for two random holes
    per row (no full rows): random placements then end the episode within a dozen transitions."""
    board, rec, rng = e.get_state()
    B, H, W = board.shape
    g = np.random.default_rng(seed)
    sel = np.arange(0, B, every)
    tall = np.full((len(sel), H - top, W), 2, np.uint8)
    holes = g.integers(0, W, (len(sel), H - top, 2))
    np.put_along_axis(tall, holes, 0, axis=2)
    board[sel, top:, :] = tall
    e.set_state(board=board, rec=rec, rng=rng)
    o.set_state(board=board, rec=rec, rng=rng)


@pytest.mark.parametrize("W,H,Q,B,grouped,ring_fill", [(20, 40, 7, 65536, True, 1), (20, 40, 7, 65536, True, 2),
                                                         (10, 20, 4, 262144, False, 1), (10, 20, 7, 131072, True, 3)])
def test_oversubscribed_grid_inline_shuffle_and_resync(W, H, Q, B, grouped, ring_fill, monkeypatch):
    """Grids larger than the chip's residency (20x40: 4 096 + 512 workgroups; 10x20 above ~87k envs) start their RNG-role
    workgroups AFTER step workgroups of the same launch have finished.  With a short look-ahead ring (TG_RING_FILL) the
    logic wave shuffles in line and the RNG role resynchronises from the state that shuffle left — a state the same
    launch's logic wave must not overwrite (round 2 kept it in tg_rng_rec: a late RNG workgroup then continued from the
    wrong frontier and the piece sequence silently left the reference's).  Fused rollouts, tall start boards so that
    episodes end and same-step resets run all the time; pieces, bags, queues and every generator against the oracle."""
    monkeypatch.setenv("TG_RING_FILL", str(ring_fill))
    monkeypatch.setenv("TG_FUSE", "8")
    e = Engine(B, W, H, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=grouped, autoreset=N.AUTORESET_SAME_STEP)
    o = po.OracleBatch(B, W, H, queue_size=Q, nthreads=8)
    seeds = np.arange(1, B + 1, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    _tall_boards(e, o, every=1, top=H // 4)
    rng = np.random.default_rng(ring_fill)
    n_done = 0
    for rnd in range(3):
        K = (24, 17, 8)[rnd]
        if grouped:
            acts, rew, term, lines = e.grouped_rollout(K, seed=9, counter0=100 * rnd)
        else:
            acts = rng.integers(0, 8, (K, B)).astype(np.int8)
            _, rew, term, lines = e.rollout(acts, want_obs=False)
        for k in range(K):
            if grouped:
                ro, to, lo, ill = o.grouped_step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
                assert not ill.any()
            else:
                ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            assert np.array_equal(rew[k], ro) and np.array_equal(term[k], to) and np.array_equal(lines[k], lo), f"round {rnd} k={k}"
        n_done += int(term.sum())
        be, re_, ge = e.get_state(); bo, ro2, go = o.get_state()
        np.testing.assert_array_equal(be, bo)
        for f in re_.dtype.names:
            if f != "reserved":
                np.testing.assert_array_equal(re_[f], ro2[f], err_msg=f"{f} round {rnd}")
        for f in ("state_lo", "state_hi", "inc_lo", "inc_hi", "uinteger", "has_uint32"):
            np.testing.assert_array_equal(ge[f], go[f], err_msg=f"rng.{f} round {rnd}")
    assert n_done > B // 2  # resets (a shuffle each, two with queue_size 7) ran in practically every launch
    e.close()
