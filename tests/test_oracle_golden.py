"""The CPU oracle (oracle/tg_oracle.c) against golden vectors generated from the reference NumPy env.
CPU-only: pins the oracle before it is trusted as the checker for the HIP engine."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay


class OracleBackend:
    def __init__(self, batch, width, height, gravity, queue_size, action_ids=None, rewards=None):
        self.o = po.OracleBatch(batch, width, height, po.RULESET_NUMPY, gravity, queue_size, action_ids=action_ids, rewards=rewards)

    def rewards_f64(self):
        return self.o.last_rewards()

    def reset(self, seeds=None, mask=None):
        self.o.reset(seeds, mask)

    def step(self, actions):
        return self.o.step(actions)

    def observe_dict(self):
        return self.o.observe(po.OBS_DICT)

    def get_state(self):
        return self.o.get_state()

    def set_state(self, **kw):
        self.o.set_state(**kw)


TRACES = ["trace_10x20_g1", "trace_10x20_g0", "trace_20x40_g1", "trace_6x12_g1", "clears_10x20", "clears_20x40",
          "trace_7x14_q2_g1", "trace_12x24_q7_g0"]  # (the last two: board sizes the engine compiles at run time)


@pytest.mark.parametrize("name", TRACES)
def test_oracle_replays_reference_trace(golden_dir, name):
    fx = replay.load(golden_dir, name)
    replay.replay_trace(OracleBackend, fx)


MAPPED = ["mapped_10x20_g1", "mapped_10x20_g0", "mapped_clears_10x20", "mapped_alias_left_10x20", "mapped_alias_swap_10x20"]


@pytest.mark.parametrize("name", MAPPED)
def test_oracle_replays_reference_trace_with_custom_mappings(golden_dir, name):
    """Tetris(actions_mapping=..., rewards_mapping=...): a permutation of the action ids with float rewards, a mapping with a
    duplicate value and ids no field holds, line clears with integer rewards — outputs of the reference itself."""
    fx = replay.load(golden_dir, name)
    replay.replay_trace(OracleBackend, fx)
    r = fx["reward_f64"]
    al, _, go, _ = fx["rewards_mapping"]
    assert (r == go).any() and (r == al).any()  # both the commit and the game-over reward occur
    if name == "mapped_clears_10x20":
        assert int(fx["lines"].max()) == 4 and {10.0, 40.0, 160.0} <= set(np.unique(r))


def test_oracle_grouped_trace_with_custom_mappings(golden_dir):
    """GroupedActionsObservations over a Tetris with non-stock mappings, terminate_on_illegal_action=False: the illegal action is
    the env's own no_op and its reward rewards.invalid_action (wrappers/grouped.py:225-239)."""
    fx = replay.load(golden_dir, "mapped_grouped_10x20")
    W, H, B, T, Q = int(fx["width"]), int(fx["height"]), int(fx["B"]), int(fx["T"]), int(fx["queue_size"])
    o = po.OracleBatch(B, W, H, queue_size=Q, action_ids=fx["action_ids"], rewards=fx["rewards_mapping"])
    o.reset(1 + np.arange(B, dtype=np.uint64))
    np.testing.assert_array_equal(o.grouped_enumerate(), fx["reset_mask"])
    n_ill = 0
    for t in range(T):
        rew, term, lines, ill = o.grouped_step(fx["actions"][t].astype(np.int32), terminate_on_illegal=bool(fx["terminate_on_illegal"]))
        np.testing.assert_array_equal(o.last_rewards(), fx["reward_f64"][t], err_msg=f"reward t={t}")
        np.testing.assert_array_equal(term, fx["done"][t], err_msg=f"t={t}")
        np.testing.assert_array_equal(lines, fx["lines"][t])
        np.testing.assert_array_equal(o.get_state()[0], replay.crop(fx["board_after"][t]), err_msg=f"board t={t}")
        n_ill += int(ill.sum())
        if term.any():
            o.reset(mask=term)
        np.testing.assert_array_equal(o.grouped_enumerate(), fx["masks"][t], err_msg=f"mask t={t}")
    assert n_ill > 50 and (fx["reward_f64"] == fx["rewards_mapping"][3]).sum() == n_ill


def test_clear_traces_really_clear_lines(golden_dir):
    fx = replay.load(golden_dir, "clears_10x20")
    assert int(fx["lines"].sum()) > 100 and int(fx["lines"].max()) == 4
    assert set(np.unique(fx["reward"])) >= {0, 1, 11, 41, 161}


def test_board_obs_is_crop_of_dict_board(golden_dir):
    o = po.OracleBatch(8)
    o.reset(seeds=np.arange(1, 9, dtype=np.uint64))
    rng = np.random.default_rng(0)
    for _ in range(50):
        o.step(rng.integers(0, 8, 8))
        d = o.observe(po.OBS_DICT)
        b = o.observe(po.OBS_BOARD)
        np.testing.assert_array_equal(b, d["board"][:, :20, 4:14])
