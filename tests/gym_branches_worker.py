"""Worker of tests/test_gym_branches.py (child process: tests/gym_shim on sys.path as `gymnasium`, TG_LIB = the CPU checker's ABI library):
executes the HAVE_GYMNASIUM branches of the package."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "gym_shim"))
sys.path.insert(0, ROOT)
import gymnasium as gym  # noqa: E402  (the look-alike)

assert gym.__version__.endswith("shim")
from tetris_gymnasium_amd import _gym_compat as G  # noqa: E402

assert G.HAVE_GYMNASIUM and G.EnvBase is gym.Env and G.VectorEnvBase is gym.vector.VectorEnv and G.spaces is gym.spaces
import tetris_gymnasium_amd.envs  # noqa: E402,F401  (registers the id)
from tetris_gymnasium_amd.envs.tetris import Tetris  # noqa: E402
from tetris_gymnasium_amd.vector import TetrisVectorEnv  # noqa: E402
from tetris_gymnasium_amd.wrappers import FeatureVectorObservation, GroupedActionsObservations, RgbObservation  # noqa: E402

# gym.make of the registered id, the way the reference's examples build theirs (envs/__init__.py:10-14)
env = gym.make("tetris_gymnasium_amd/Tetris", width=8, height=16)
assert isinstance(env, gym.Env) and isinstance(env.unwrapped, Tetris) and env.unwrapped.width == 8
obs, info = env.reset(seed=1)
assert env.observation_space.contains(obs), {k: (v.shape, v.dtype) for k, v in obs.items()}
for t in range(40):
    a = env.action_space.sample()
    assert env.action_space.contains(a)
    obs, r, term, trunc, info = env.step(a)
    assert env.observation_space.contains(obs) and trunc is False and "lines_cleared" in info
    if term:
        obs, info = env.reset()
rgb = RgbObservation(env)
assert rgb.observation_space.contains(rgb.observation(None))
grp = GroupedActionsObservations(env, observation_wrappers=[FeatureVectorObservation(env)])
g, info = grp.reset(seed=2)
assert g.shape == grp.observation_space.shape and info["action_mask"].shape == (grp.action_space.n,)
grp.close()

for mode, enum in (("next_step", gym.vector.AutoresetMode.NEXT_STEP), ("same_step", gym.vector.AutoresetMode.SAME_STEP)):
    v = TetrisVectorEnv(32, observation="dict", autoreset_mode=mode)
    assert isinstance(v, gym.vector.VectorEnv) and v.metadata["autoreset_mode"] is enum
    assert isinstance(v.action_space, gym.spaces.MultiDiscrete) and v.action_space.shape == (32,)
    obs, info = v.reset(seed=3)
    assert v.observation_space.contains(obs) and v.single_observation_space.contains({k: o[0] for k, o in obs.items()})
    n = 0
    for t in range(120):
        a = v.action_space.sample()
        obs, rew, term, trunc, infos = v.step(a)
        assert v.observation_space.contains(obs) and rew.shape == (32,) and term.dtype == np.bool_ and trunc.dtype == np.bool_
        if "episode" in infos:
            m = infos["_episode"]
            assert {"r", "l", "t"} <= set(infos["episode"]) and (infos["episode"]["l"][m] > 0).all()
            n += int(m.sum())
    assert n > 0 and len(v.get_attr("width")) == 32 and v.call("height")[0] == 20
    v.close()
b = TetrisVectorEnv(8, observation="board", grouped=True)
assert isinstance(b.observation_space, gym.spaces.Box) and b.observation_space.shape == (8, 20, 10) and b.action_space.nvec[0] == 40
b.close()
print("GYM_BRANCHES_OK")
