"""CPU tests of the host-side mirror of the reference interface (no GPU, no engine calls)."""
import numpy as np
import pytest

from tetris_gymnasium_amd.components.tetromino import identify, make_tetromino
from tetris_gymnasium_amd.envs import tetris_fn as F
from tetris_gymnasium_amd.functional import TETROMINOES, EnvConfig, State
from tetris_gymnasium_amd.functional.core import create_board, get_initial_x_y
from tetris_gymnasium_amd.mappings.actions import ActionsMapping
from tetris_gymnasium_amd.mappings.rewards import RewardsMapping
from tetris_gymnasium_amd.vector import default_seeds, shard_ranges


def test_mappings_match_reference_constants():  # mappings/actions.py:12-19, rewards.py:12-15
    a, r = ActionsMapping(), RewardsMapping()
    assert (a.move_left, a.move_right, a.move_down, a.rotate_clockwise, a.rotate_counterclockwise, a.hard_drop, a.swap,
            a.no_op) == tuple(range(8))
    assert (r.alife, r.clear_line, r.game_over, r.invalid_action) == (1, 1, 0, -0.1)


def test_tetrominoes_lut():  # tests/test_functional/test_tetrominoes.py:13-19
    assert TETROMINOES.matrices.shape == (7, 4, 4, 4) and TETROMINOES.matrices.dtype == np.int8
    assert list(TETROMINOES.ids) == [2, 3, 4, 5, 6, 7, 8]
    assert (TETROMINOES.matrices.sum(axis=(2, 3)) == 4).all()
    assert TETROMINOES.colors.shape == (7, 3)


def test_board_and_spawn():  # tests/test_functional/test_core/test_board.py:7-44
    cfg = EnvConfig(10, 20, 4, 7, True)
    b = create_board(cfg)
    assert b.shape == (24, 18) and (b[20:] == 1).all() and (b[:20, 4:14] == 0).all() and (b[:, :4] == 1).all()
    assert get_initial_x_y(cfg) == (7, 0)


def test_identify_roundtrip():
    for p in range(7):
        for r in range(4):
            for v in (1, p + 2):
                assert identify(make_tetromino(p, r, v))[2] == v
                ip, ir, _ = identify(make_tetromino(p, r, v))
                np.testing.assert_array_equal(make_tetromino(ip, ir, v).matrix, make_tetromino(p, r, v).matrix)
    with pytest.raises(ValueError):
        from tetris_gymnasium_amd.components.tetromino import Tetromino
        identify(Tetromino(2, [0, 0, 0], np.array([[1, 1, 1], [1, 0, 1], [0, 0, 0]], np.uint8)))


def test_threefry_and_split_published_values():
    # Random123 known-answer vectors (also in jax's own test-suite)
    assert F.threefry2x32((0, 0), (0, 0)) == (0x6B200159, 0x99BA4EFE)
    assert F.threefry2x32((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF)) == (0x1CB996FC, 0xBB002BE7)
    assert F.threefry2x32((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3)) == (0xC4923A9C, 0x483DF7A0)
    assert list(F.PRNGKey(42)) == [0, 42]
    assert F.split(F.PRNGKey(42)).shape == (2, 2)


def test_original_split_scheme_matches_values_published_in_jax_docs():
    """jax's pre-0.5 default (non-partitionable) split of PRNGKey(42) / PRNGKey(0), as printed in the jax
    documentation; the oracle's restatement of that scheme reproduces them (pins threefry + counter layout)."""
    import ctypes as C
    from oracle import pyoracle as po
    L = po.lib()
    L.tgo_set_threefry_partitionable(0)
    try:
        for key, exp in (((0, 42), ([2465931498, 3679230171], [255383827, 267815257])),
                         ((0, 0), ([4146024105, 967050713], [2718843009, 1272950319]))):
            k = np.array(key, np.uint32); a = np.zeros(2, np.uint32); b = np.zeros(2, np.uint32)
            L.tgo_jax_split(k.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
            assert (a.tolist(), b.tolist()) == exp
    finally:
        L.tgo_set_threefry_partitionable(1)
    # host split (partitionable, jax >= 0.5 default) agrees with the oracle's
    k = np.array([0, 42], np.uint32); a = np.zeros(2, np.uint32); b = np.zeros(2, np.uint32)
    L.tgo_jax_split(k.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    assert F.split(k).tolist() == [a.tolist(), b.tolist()]


def test_state_replace_drops_device_token():  # tests/test_functional/test_env/test_step.py:19 uses state.replace
    s = State(rng_key=np.zeros(2, np.uint32), board=create_board(EnvConfig(10, 20, 4, 7)), active_tetromino=np.int32(1),
              rotation=np.int32(0), x=np.int32(7), y=np.int32(0), queue=np.arange(7, dtype=np.int32),
              queue_index=np.int32(1), game_over=np.bool_(False), score=np.float32(0), _token=(1, 5))
    s2 = s.replace(game_over=np.bool_(True))
    assert s2._token is None and bool(s2.game_over) and not bool(s.game_over) and s._token == (1, 5)


def test_rejects_custom_plug_points_before_touching_the_gpu():
    cfg = EnvConfig(10, 20, 4, 7)
    with pytest.raises(NotImplementedError):
        F.reset(TETROMINOES, F.PRNGKey(0), cfg, create_queue_fn=lambda c, k: None)
    with pytest.raises(NotImplementedError):
        F.reset(TETROMINOES, F.PRNGKey(0), EnvConfig(10, 20, 2, 7))  # (padding >= 3 is served: test_gpu_surfaces.py)
    with pytest.raises(NotImplementedError):
        F.reset(object(), F.PRNGKey(0), cfg)


def test_get_observation_host_restatement():
    cfg = EnvConfig(10, 20, 4, 7)
    b = create_board(cfg)
    b[19, 4:9] = 3
    o = F.get_observation(b, 7, 0, 2, 0, False, TETROMINOES, cfg)
    assert o.shape == (20, 10) and (o == -1).sum() == 4 and (o == 1).sum() == 5
    assert (F.get_observation(b, 7, 0, 2, 0, True, TETROMINOES, cfg) == -1).sum() == 0


def test_shard_ranges_and_seeds():
    assert shard_ranges(524288, 8) == [(i * 65536, (i + 1) * 65536) for i in range(8)]
    r = shard_ranges(10, 4)
    assert r == [(0, 3), (3, 6), (6, 8), (8, 10)] and sum(h - l for l, h in r) == 10
    assert list(default_seeds(3, 6)) == [4, 5, 6] and list(default_seeds(0, 2, seed=100)) == [100, 101]
