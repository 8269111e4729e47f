"""Oracle RgbObservation surface (wrappers/observation.py:38-74) against frames the reference produced along random
traces with hold swaps: board | queue strip (top right) | holder (bottom right) through the pixel colour table."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay


@pytest.mark.parametrize("name", ["rgb_10x20", "rgb_20x40"])
def test_rgb_frames(golden_dir, name):
    fx = replay.load(golden_dir, name)
    W, H, Q, B, T = (int(fx[k]) for k in ("width", "height", "queue_size", "B", "T"))
    o = po.OracleBatch(B, W, H, queue_size=Q)
    o.reset(seeds=1 + np.arange(B, dtype=np.uint64))
    np.testing.assert_array_equal(o.observe(po.OBS_RGB), fx["first"])
    for t in range(T):
        _, term, _ = o.step(fx["actions"][t].astype(np.int32))
        np.testing.assert_array_equal(term, fx["done"][t], err_msg=f"t={t}")
        np.testing.assert_array_equal(o.observe(po.OBS_RGB), fx["frames"][t], err_msg=f"frame t={t}")
        if term.any():
            o.reset(mask=term)
            got = o.observe(po.OBS_RGB)
            m = term.astype(bool)
            np.testing.assert_array_equal(got[m], fx["reset_frames"][t][m], err_msg=f"reset frame t={t}")
    assert fx["done"].sum() > 0 and (fx["actions"] == 6).sum() > 50  # episodes ended, many hold swaps
