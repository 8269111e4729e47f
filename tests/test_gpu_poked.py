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
    m = replay.replay_poked(be, fx, skip=fx["bedrock_row"].astype(bool), upload=_dev_upload(be.e) if how == "device" else None)
    assert m.sum() >= B - 8 * int(fx["bedrock_row"].sum())
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
    replay.replay_poked(Be, fx, skip=fx["bedrock_row"].astype(bool), upload=_dev_upload(e), check_obs=False, rollout=ro)
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
    skip = fx["bedrock_row"].astype(bool)
    replay.poked_upload(fx, board, rec, skip)
    e.set_state(board=board, rec=rec)
    ng = fx["g_mask"].shape[0]
    ok = ~skip
    mask, boards = e.placement_boards()
    np.testing.assert_array_equal(mask[::8][:ng][ok[:ng]], fx["g_mask"][ok[:ng]])
    np.testing.assert_array_equal(boards[::8][:ng][ok[:ng]], fx["g_boards"][ok[:ng]])
    pl, cu = e.features_as(placement=True, current=True)
    np.testing.assert_array_equal(pl[::8][:ng][ok[:ng]], fx["g_feats"][ok[:ng]])
    np.testing.assert_array_equal(cu[::8][ok], fx["cur_feats"][ok])
    np.testing.assert_array_equal(e.observe_as(N.OBS_RGB)[::8][ok], fx["rgb"][ok])
    e.close()


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("name", ["poked_10x20", "poked_20x40"])
def test_engine_grouped_steps_from_poked_states(golden_dir, name, packed, monkeypatch):
    """Every one of the grouped wrapper's 4W actions from every hand-made state through the step kernel's grouped action set
    (wrappers/grouped.py:209-269): reward (-0.1 on a masked-out action), termination, lines, board / piece / position after."""
    monkeypatch.setenv("TG_PACKED", "1" if packed else "0")
    fx = replay.load(golden_dir, name)
    ng, NA = fx["gs_reward"].shape
    e = Engine(ng * NA, int(fx["width"]), int(fx["height"]), ruleset=N.RULESET_NUMPY, queue_size=int(fx["queue_size"]), obs_mode=N.OBS_BOARD,
               grouped=True, terminate_on_illegal=True)
    e.reset(replay.poked_seeds(fx, NA, ng))
    replay.replay_poked_grouped(e.step, e.get_state, lambda b, r: e.set_state(board=b, rec=r), fx, skip=fx["bedrock_row"].astype(bool))
    e.close()


def test_bedrock_rows_inside_the_playfield_are_refused(golden_dir):
    """A playfield row of bedrock cells (value 1 in every column) is never taken by the reference's clear_filled_rows
    (envs/tetris.py:493: `~(board == 1).all(axis=1)`); the engine's row test is `every cell non-zero`, so such a board is refused
    outright (TG_EUNSUPPORTED) rather than stepped differently.  The oracle replays those states (tests/test_oracle_poked.py)."""
    fx = replay.load(golden_dir, "poked_10x20")
    i = int(np.flatnonzero(fx["bedrock_row"])[0])
    e = Engine(8, 10, 20, obs_mode=N.OBS_BOARD)
    e.reset(np.arange(1, 9, dtype=np.uint64))
    b, r, g = e.get_state()
    b0 = b.copy()
    b[3] = replay.crop(fx["st_board"])[i]
    with pytest.raises(Exception, match="bedrock"):
        e.set_state(board=b)
    np.testing.assert_array_equal(e.get_state()[0], b0)  # (refused in front of the upload: the engine is as it was)
    # ... and through tg_set_state_dev: the host cannot look, the kernel that rebuilds the column masks flags it and the next call
    # that waits for the stream reports it once (TG_EUNSUPPORTED)
    db, dr, dg = e.get_state_device()
    e.dev_upload(db.ptr, np.ascontiguousarray(b))
    e.set_state_device(board=db)
    with pytest.raises(Exception, match="bedrock"):
        e.sync()
    e.set_state(board=np.zeros_like(b))  # a clean board again: the handle goes on
    e.sync()
    e.step(np.zeros(8, np.int32))
    e.close()
