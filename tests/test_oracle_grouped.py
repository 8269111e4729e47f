"""Oracle grouped-action surface (wrappers/grouped.py) against reference-generated fixtures, including the
reference's only golden file (expected_result_i_placement.csv) and its 40-entry legal mask."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import replay


def _state_from_matrix(o, padded_board, piece, matrix16):
    """Put env 0.. of an OracleBatch into (board, active piece with the given 4x4 matrix)."""
    B = padded_board.shape[0]
    board, rec, rng = o.get_state()
    for i in range(B):
        val = int(matrix16[i].max())
        rot = next(r for r in range(4) if np.array_equal(replay.matrix16(piece[i], r, val), matrix16[i]))
        rec["piece"][i], rec["rot"][i], rec["val"][i] = piece[i], rot, val
    o.set_state(board=replay.crop(padded_board), rec=rec)


def test_reference_csv_and_mask(golden_dir):
    fx = replay.load(golden_dir, "ref_tests_grouped")
    o = po.OracleBatch(1)
    o.reset(seeds=np.array([42], np.uint64))
    vi = np.rot90(replay.NATIVE[0])  # vertical I with matrix values 1 (tests/conftest.py:37-42)
    m16 = np.zeros((4, 4), np.uint8); m16[:4, :4] = vi
    _state_from_matrix(o, fx["mock_board"][None], np.array([0]), m16.reshape(1, 16))
    mask, boards = o.grouped_enumerate(boards=True)
    np.testing.assert_array_equal(mask[0], fx["mask"])
    np.testing.assert_array_equal(boards[0], fx["boards40"])
    rew, term, lines, ill = o.grouped_step(np.array([5 * 4 + 1]))
    np.testing.assert_array_equal(o.padded_boards()[0], fx["csv_board"])
    assert not ill[0]


@pytest.mark.parametrize("name", ["grouped_10x20", "grouped_20x40", "grouped_clears_10x20"])
def test_grouped_trace(golden_dir, name):
    fx = replay.load(golden_dir, name)
    W, H, B, T, Q = int(fx["width"]), int(fx["height"]), int(fx["B"]), int(fx["T"]), int(fx["queue_size"])
    init = fx["init_boards"] if "init_boards" in fx.files and fx["init_boards"].size else None
    o = po.OracleBatch(B, W, H, queue_size=Q)
    o.reset(seeds=1 + np.arange(B, dtype=np.uint64))
    if init is not None:
        o.set_state(board=replay.crop(init))
    np.testing.assert_array_equal(o.grouped_enumerate(), fx["reset_mask"])
    for t in range(T):
        rew, term, lines, ill = o.grouped_step(fx["actions"][t].astype(np.int32))
        np.testing.assert_allclose(rew, fx["reward"][t], rtol=0, atol=0, err_msg=f"t={t}")
        np.testing.assert_array_equal(term, fx["done"][t], err_msg=f"t={t}")
        np.testing.assert_array_equal(lines, fx["lines"][t])
        np.testing.assert_array_equal(o.padded_boards(), fx["board_after"][t], err_msg=f"t={t}")
        if term.any():
            o.reset(mask=term)
            if init is not None:
                b, _, _ = o.get_state()
                b[term.astype(bool)] = replay.crop(init)[term.astype(bool)]
                o.set_state(board=b)
        np.testing.assert_array_equal(o.grouped_enumerate(), fx["masks"][t], err_msg=f"mask t={t}")
    assert fx["lines"].sum() > 0 or "clears" not in name


@pytest.mark.parametrize("name", ["grouped_10x20", "grouped_20x40", "grouped_clears_10x20"])
def test_grouped_enumeration_boards(golden_dir, name):
    fx = replay.load(golden_dir, name)
    n = fx["enum_mask"].shape[0]
    assert n > 0
    o = po.OracleBatch(n, int(fx["width"]), int(fx["height"]), queue_size=int(fx["queue_size"]))
    o.reset(seeds=1 + np.arange(n, dtype=np.uint64))
    _state_from_matrix(o, fx["enum_state_board"], fx["enum_piece"], fx["enum_matrix"])
    mask, boards = o.grouped_enumerate(boards=True)
    np.testing.assert_array_equal(mask, fx["enum_mask"])
    np.testing.assert_array_equal(boards, fx["enum_boards"])


def test_feature_vector(golden_dir):
    fx = replay.load(golden_dir, "features")
    for b, m, f in zip(fx["boards"], fx["masks"], fx["feats"]):
        np.testing.assert_array_equal(po.feature_vector(b, m, 10, 20), f)
    g = replay.load(golden_dir, "ref_tests_grouped")
    # tests/helpers/mock.py:35-39 known answers (mask all-zero → only row 0 is wiped)
    f = po.feature_vector(g["mock_board"], np.zeros_like(g["mock_board"]), 10, 20)
    assert list(f[:10]) == list(g["heights"]) and f[10] == 11 and f[11] == 3 and f[12] == 14


@pytest.mark.parametrize("name", ["grouped_features_10x20", "grouped_features_20x40", "grouped_features_prefull_6x12",
                                  "grouped_features_prefull_10x20"])
def test_grouped_placement_features(golden_dir, name):
    """What examples/train_lin_grouped.py trains on: [4W][W+3] feature tensor per state + info["board"]."""
    fx = replay.load(golden_dir, name)
    n = fx["boards"].shape[0]
    o = po.OracleBatch(n, int(fx["width"]), int(fx["height"]), queue_size=int(fx["queue_size"]))
    o.reset(seeds=1 + np.arange(n, dtype=np.uint64))
    _state_from_matrix(o, fx["boards"], fx["pieces"], fx["mats"])
    _, rec, _ = o.get_state()
    rec["x"] = (int(fx["width"]) + 8) // 2 - np.where(fx["pieces"] == 0, 2, 1)  # states were recorded right after a spawn
    rec["y"] = 0
    o.set_state(rec=rec)
    np.testing.assert_array_equal(o.grouped_enumerate(), fx["masks"])
    np.testing.assert_array_equal(o.grouped_features(), fx["feats"])
    np.testing.assert_array_equal(o.features(), fx["cur_feats"])
    if "pboards" in fx:  # (the *_prefull fixtures: boards with rows that were full BEFORE the placement — every placement clears them too)
        np.testing.assert_array_equal(o.grouped_enumerate(boards=True)[1], fx["pboards"])
