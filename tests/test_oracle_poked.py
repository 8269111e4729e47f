"""The CPU oracle against HAND-MADE states stepped by the reference itself (tests/golden/poked_*.npz, oracle/refgen/make_golden.py
gen_poked): boards with rows already full, fills up to the spawn rows, a playfield row of bedrock cells, the piece anywhere the
reference's slicing tolerates (on filled cells, inside the bedrock), arbitrary holder / queue / bag / has_swapped / game_over — the
state class of the reference's own tests (tests/test_base_env/**) and the one where round 5's two late bugs lived.  CPU-only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay
from tests.test_oracle_golden import OracleBackend

POKED = ["poked_10x20", "poked_10x20_g0", "poked_20x40"]


@pytest.mark.parametrize("name", POKED)
def test_oracle_replays_poked_states(golden_dir, name):
    fx = replay.load(golden_dir, name)
    be = OracleBackend(8 * int(fx["n_states"]), int(fx["width"]), int(fx["height"]), bool(fx["gravity"]), int(fx["queue_size"]))
    be.reset(seeds=replay.poked_seeds(fx))
    replay.replay_poked(be, fx)
    assert int(fx["bedrock_row"].sum()) > 0 and int(fx["st_game_over"].sum()) > 0 and int(fx["lines"][0].sum()) > 50


@pytest.mark.parametrize("name", ["poked_10x20", "poked_20x40"])
def test_oracle_wrappers_on_poked_states(golden_dir, name):
    """The grouped wrapper's legal mask / placement boards / per-placement features, FeatureVectorObservation and RgbObservation of the
    same hand-made states (wrappers/grouped.py:124-207, wrappers/observation.py:38-74, 236-278)."""
    fx = replay.load(golden_dir, name)
    n, W, H, Q = int(fx["n_states"]), int(fx["width"]), int(fx["height"]), int(fx["queue_size"])
    o = po.OracleBatch(8 * n, W, H, po.RULESET_NUMPY, True, Q)
    o.reset(replay.poked_seeds(fx))
    board, rec, _ = o.get_state()
    replay.poked_upload(fx, board, rec)
    o.set_state(board=board, rec=rec)
    ng = fx["g_mask"].shape[0]
    mask, boards = o.grouped_enumerate(boards=True)
    np.testing.assert_array_equal(mask[::8][:ng], fx["g_mask"])
    np.testing.assert_array_equal(boards[::8][:ng], fx["g_boards"])
    np.testing.assert_array_equal(o.grouped_features()[::8][:ng], fx["g_feats"])
    np.testing.assert_array_equal(o.features()[::8], fx["cur_feats"])
    np.testing.assert_array_equal(o.observe(po.OBS_RGB)[::8], fx["rgb"])


@pytest.mark.parametrize("name", ["poked_10x20", "poked_20x40"])  # (the gravity-off job carries no grouped part)
def test_oracle_grouped_steps_from_poked_states(golden_dir, name):
    """Every grouped action (column x rotation: teleport + hard drop, illegal -> terminate with -0.1) from every hand-made state."""
    fx = replay.load(golden_dir, name)
    ng, NA = fx["gs_reward"].shape
    o = po.OracleBatch(ng * NA, int(fx["width"]), int(fx["height"]), po.RULESET_NUMPY, bool(fx["gravity"]), int(fx["queue_size"]))
    o.reset(replay.poked_seeds(fx, NA, ng))
    replay.replay_poked_grouped(lambda a: o.grouped_step(a, terminate_on_illegal=True)[:3], o.get_state,
                                lambda b, r: o.set_state(board=b, rec=r), fx)
