"""The HIP engine on HAND-MADE states stepped by the reference itself (tests/golden/poked_*.npz): the state class of
the reference's own tests (tests/test_base_env/** assign board / x / y / active_tetromino / holder / queue / flags directly) and the one
where round 5's two late bugs lived.  Uploaded through tg_set_state AND tg_set_state_dev, stepped one and sixteen transitions per launch,
with byte and nibble-packed boards, and seen through the grouped / feature / RGB surfaces."""
import numpy as np
import pytest

from tests import replay

pytestmark = pytest.mark.gpu

from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402
from tests.test_gpu_parity import EngineBackend, EngineInStepObsBackend  # noqa: E402

POKED = ["poked_10x20", "poked_10x20_g0", "poked_20x40"]


def _geo(fx):
    return 8 * int(fx["n_states"]), int(fx["width"]), int(fx["height"]), bool(fx["gravity"]), int(fx["queue_size"])


def _dev_upload(e):
    """(board, rec) -> tg_set_state_dev: the arrays go into the engine's own snapshot buffers on the device first"""
    def up(board, rec):
        db, dr, dg = e.get_state_device()
        e.dev_upload(db.ptr, np.ascontiguousarray(board)); e.dev_upload(dr.ptr, np.ascontiguousarray(rec))
        e.set_state_device(board=db, rec=dr)
    return up


@pytest.mark.parametrize("how", ["host", "device"])
@pytest.mark.parametrize("name", POKED)
def test_engine_replays_poked_states(golden_dir, name, how):
    """Every output, every observation checksum (the observe kernel's re-pack behind the upload, the step kernel's own pack behind every
    transition), position / piece after every transition, the full state behind the first transition and at the end."""
    fx = replay.load(golden_dir, name)
    B, W, H, g, Q = _geo(fx)
    be = (EngineBackend if how == "host" else EngineInStepObsBackend)(B, W, H, g, Q)
    be.reset(seeds=replay.poked_seeds(fx))
    m = replay.replay_poked(be, fx, upload=_dev_upload(be.e) if how == "device" else None)
    assert m.all() and int(fx["bedrock_row"].sum()) > 0  # (incl. the states with a playfield row of bedrock cells)
    be.e.close()


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", ["poked_10x20", "poked_20x40"])
def test_engine_replays_poked_states_board_handles(golden_dir, name, fused, packed, monkeypatch):
    """The same states on TG_OBS_BOARD handles — the instantiations the benchmarks run —, boards as bytes and as nibbles (TG_PACKED=1),
    one transition per launch and all sixteen in one tg_rollout launch, uploaded through tg_set_state_dev."""
    monkeypatch.setenv("TG_PACKED", "1" if packed else "0")
    fx = replay.load(golden_dir, name)
    B, W, H, g, Q = _geo(fx)
    e = Engine(B, W, H, ruleset=N.RULESET_NUMPY, gravity=g, queue_size=Q, obs_mode=N.OBS_BOARD, fuse=16)
    assert bool(int(e.step_kernel_name().rstrip(">").split(",")[-1]) & 256) == packed
    e.reset(replay.poked_seeds(fx))

    class Be:
        get_state = staticmethod(e.get_state)
        step = staticmethod(e.step)
    ro = None
    if fused:
        def ro(acts):
            res = e.rollout(acts, want_obs=False)
            return res[1], res[2], res[3]
    replay.replay_poked(Be, fx, upload=_dev_upload(e), check_obs=False, rollout=ro)
    e.close()


@pytest.mark.parametrize("name", ["poked_10x20", "poked_20x40"])
def test_engine_wrapper_surfaces_on_poked_states(golden_dir, name):
    """wrappers/grouped.py:124-207 and wrappers/observation.py:38-74, 236-278 on the hand-made states: legal mask, the 4W placement
    boards, the per-placement and current feature vectors, the RGB frame — rows already full, pieces on filled cells and all."""
    fx = replay.load(golden_dir, name)
    B, W, H, g, Q = _geo(fx)
    e = Engine(B, W, H, ruleset=N.RULESET_NUMPY, gravity=g, queue_size=Q, obs_mode=N.OBS_BOARD, grouped=True)
    e.reset(replay.poked_seeds(fx))
    board, rec, _ = e.get_state()
    replay.poked_upload(fx, board, rec)
    e.set_state(board=board, rec=rec)
    ng = fx["g_mask"].shape[0]
    mask, boards = e.placement_boards()
    np.testing.assert_array_equal(mask[::8][:ng], fx["g_mask"])
    np.testing.assert_array_equal(boards[::8][:ng], fx["g_boards"])
    pl, cu = e.features_as(placement=True, current=True)
    np.testing.assert_array_equal(pl[::8][:ng], fx["g_feats"])
    np.testing.assert_array_equal(cu[::8], fx["cur_feats"])
    np.testing.assert_array_equal(e.observe_as(N.OBS_RGB)[::8], fx["rgb"])
    e.close()


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("name", ["poked_10x20", "poked_20x40"])  # (the gravity-off job carries no grouped part)
def test_engine_grouped_steps_from_poked_states(golden_dir, name, packed, monkeypatch):
    """Every one of the grouped wrapper's 4W actions from every hand-made state through the step kernel's grouped action set
    (wrappers/grouped.py:209-269): reward (-0.1 on a masked-out action), termination, lines, board / piece / position after."""
    monkeypatch.setenv("TG_PACKED", "1" if packed else "0")
    fx = replay.load(golden_dir, name)
    ng, NA = fx["gs_reward"].shape
    e = Engine(ng * NA, int(fx["width"]), int(fx["height"]), ruleset=N.RULESET_NUMPY, gravity=bool(fx["gravity"]), queue_size=int(fx["queue_size"]),
               obs_mode=N.OBS_BOARD, grouped=True, terminate_on_illegal=True)
    e.reset(replay.poked_seeds(fx, NA, ng))
    replay.replay_poked_grouped(e.step, e.get_state, lambda b, r: e.set_state(board=b, rec=r), fx)
    e.close()


def test_bedrock_rows_inside_the_playfield_are_never_cleared(golden_dir):
    """A playfield row of bedrock cells (value 1 in every column) is never taken by the reference's clear_filled_rows (envs/tetris.py:493:
    `& ~(board == 1).all(axis=1)`): it collides like any filled row, survives every lock, and moves down with the rows above a cleared one.
    Rounds 1-5 would have cleared it with the next lock (the column masks hold one bit per cell); now the step kernel keeps the mask of such
    rows in the two record words the NUMPY rule-set does not use.  The reference-generated states are replayed above; here: the mask is
    device-private (tg_get_state returns the words as 0), follows uploads through both paths, masked resets and fused launches."""
    from oracle import pyoracle as po
    B, W, H = 256, 10, 20
    for packed in ("0", "1"):
        import os
        os.environ["TG_PACKED"] = packed
        try:
            e = Engine(B, W, H, obs_mode=N.OBS_BOARD, autoreset=N.AUTORESET_SAME_STEP)
        finally:
            os.environ.pop("TG_PACKED", None)
        o = po.OracleBatch(B, W, H)
        seeds = np.arange(1, B + 1, dtype=np.uint64)
        e.reset(seeds); o.reset(seeds)
        rng = np.random.default_rng(int(packed))
        b, r, g = o.get_state()
        for i in range(B):  # a bedrock row low in the board, under it a row that is already full (the next lock takes it: the bedrock row
            rr = int(rng.integers(12, H - 1))  # moves down), above it a row with one hole (play completes it) and random cells
            b[i, rr - 3:rr - 1] = np.where(rng.random((2, W)) < 0.4, rng.integers(2, 9, (2, W)), 0)
            b[i, rr - 1] = 3; b[i, rr - 1, i % W] = 0
            b[i, rr] = 1
            if i % 2:
                b[i, rr + 1] = rng.integers(2, 9, W)
            if i % 5 == 0 and rr + 2 < H:
                b[i, rr + 2] = 1  # a second bedrock row
        o.set_state(board=b)
        if packed == "0":
            e.set_state(board=b)
        else:
            db, dr, dg = e.get_state_device()
            e.dev_upload(db.ptr, np.ascontiguousarray(b)); e.set_state_device(board=db)
        ge = e.get_state()[1]
        assert not ge["score"].any() and not ge["reserved"].any()
        n_lines = 0
        for t in range(30):
            a = rng.integers(0, 8, B).astype(np.int32)
            a[rng.random(B) < 0.35] = 5
            got, exp = e.step(a), o.step(a, autoreset=po.AUTORESET_SAME_STEP)
            for x, y, nm in zip(got, exp, ("reward", "terminated", "lines")):
                np.testing.assert_array_equal(x, y, err_msg=f"{nm} t={t} packed={packed}")
            n_lines += int(exp[2].sum())
            np.testing.assert_array_equal(e.observation(), o.observe(po.OBS_BOARD), err_msg=f"obs t={t}")
            if t == 7:  # (before the stacks top out and same-step autoreset wipes the boards)
                bo = o.get_state()[0]
                np.testing.assert_array_equal(e.get_state()[0], bo)
                still = int(((bo == 1).all(axis=2)).any(axis=1).sum())
                # rows were cleared (pre-full ones under a bedrock row) and the bedrock rows are still there, some of them one row lower
                assert n_lines >= B // 8 and still > B // 2, (n_lines, still)
        acts = rng.integers(0, 8, (48, B)).astype(np.int8); acts[rng.random((48, B)) < 0.35] = 5
        res = e.rollout(acts)
        for k in range(48):
            ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
            np.testing.assert_array_equal(res[1][k], ro); np.testing.assert_array_equal(res[3][k], lo)
            np.testing.assert_array_equal(res[0][k], o.observe(po.OBS_BOARD), err_msg=f"traj obs k={k}")
            n_lines += int(lo.sum())
        m = rng.random(B) < 0.3
        e.reset(mask=m); o.reset(mask=m)
        for t in range(10):
            a = np.full(B, 5, np.int32)
            got, exp = e.step(a), o.step(a, autoreset=po.AUTORESET_SAME_STEP)
            np.testing.assert_array_equal(got[2], exp[2]); np.testing.assert_array_equal(got[1], exp[1])
        np.testing.assert_array_equal(e.get_state()[0], o.get_state()[0])
        # one shape stays refused, on both upload paths: a bedrock row INSIDE the four spawn rows (the grouped wrapper's legal mask would
        # then depend on the reference's drop loop tunnelling through it: wrappers/grouped.py:147-166)
        b0 = e.get_state()[0]
        bad = b0.copy(); bad[5, 2] = 1
        with pytest.raises(Exception, match="spawn rows"):
            e.set_state(board=bad)
        np.testing.assert_array_equal(e.get_state()[0], b0)  # (refused in front of the upload: the engine is as it was)
        db, dr, dg = e.get_state_device()
        e.dev_upload(db.ptr, np.ascontiguousarray(bad)); e.set_state_device(board=db)
        with pytest.raises(Exception, match="spawn rows"):
            e.sync()
        e.set_state(board=b0)
        e.step(np.zeros(B, np.int32))
        e.close()
