"""Conformance of the adapter classes with a REAL gymnasium (>= 1.0).  gymnasium is not installed in either image of this
project, so everything here is skipped there (and the HAVE_GYMNASIUM branches of _gym_compat.py / vector.py have never
executed: INTEGRATION.md says "unverified").  Wherever `import gymnasium` works and a GPU is present this file is the check:
space / dtype conformance of `Tetris` (gymnasium.utils.env_checker.check_env's data checks, as the reference's
tests/test_integration/test_gameplay.py:25-32 runs them), `metadata["autoreset_mode"]` as the AutoresetMode enum, and a
gymnasium.wrappers.vector.RecordEpisodeStatistics wrap of TetrisVectorEnv as examples/train_ppo.py:185-190 builds it."""
import numpy as np
import pytest

gym = pytest.importorskip("gymnasium")
pytestmark = pytest.mark.gpu


def test_tetris_passes_the_env_checker():
    from gymnasium.utils.env_checker import check_env

    from tetris_gymnasium_amd.envs.tetris import Tetris

    env = Tetris()
    check_env(env, skip_render_check=True)
    env.close()


@pytest.mark.parametrize("mode", ["next_step", "same_step"])
def test_vector_env_under_record_episode_statistics(mode):
    from tetris_gymnasium_amd.vector import TetrisVectorEnv

    envs = TetrisVectorEnv(64, observation="dict", autoreset_mode=mode)
    assert isinstance(envs, gym.vector.VectorEnv)
    if hasattr(gym.vector, "AutoresetMode"):
        assert isinstance(envs.metadata["autoreset_mode"], gym.vector.AutoresetMode)
    wrapped = gym.wrappers.vector.RecordEpisodeStatistics(envs)
    obs, info = wrapped.reset(seed=1)
    assert envs.single_observation_space.contains({k: v[0] for k, v in obs.items()})
    n = 0
    for t in range(300):
        a = envs.action_space.sample()
        obs, rew, term, trunc, info = wrapped.step(a)
        assert rew.shape == (64,) and term.dtype == np.bool_ and trunc.dtype == np.bool_
        if "episode" in info:
            m = info["_episode"]
            assert (info["episode"]["l"][m] > 0).all() and {"r", "l", "t"} <= set(info["episode"])
            n += int(m.sum())
    assert n > 0
    assert len(envs.get_attr("width")) == 64 and envs.call("height")[0] == 20
    wrapped.close()


def test_gym_make_builds_the_registered_drop_in():
    """the reference's examples start with gym.make("tetris_gymnasium/Tetris", ...) (envs/__init__.py:10-14)"""
    import tetris_gymnasium_amd.envs  # noqa: F401  (registers the id)
    from tetris_gymnasium_amd.envs.tetris import Tetris

    env = gym.make("tetris_gymnasium_amd/Tetris", width=8, height=16)
    assert isinstance(env.unwrapped, Tetris) and env.unwrapped.width == 8
    obs, info = env.reset(seed=1)
    assert env.observation_space.contains(obs)
    env.close()

