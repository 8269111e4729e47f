"""The CPU oracle (oracle/tg_oracle.c) against golden vectors generated from the reference NumPy env.
CPU-only: pins the oracle before it is trusted as the checker for the HIP engine."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay


class OracleBackend:
    def __init__(self, batch, width, height, gravity, queue_size):
        self.o = po.OracleBatch(batch, width, height, po.RULESET_NUMPY, gravity, queue_size)

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


TRACES = ["trace_10x20_g1", "trace_10x20_g0", "trace_20x40_g1", "trace_6x12_g1", "clears_10x20", "clears_20x40"]


@pytest.mark.parametrize("name", TRACES)
def test_oracle_replays_reference_trace(golden_dir, name):
    fx = replay.load(golden_dir, name)
    replay.replay_trace(OracleBackend, fx)


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
