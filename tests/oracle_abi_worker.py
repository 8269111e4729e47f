"""Worker of tests/test_oracle_abi.py: runs in a child process whose TG_LIB points at oracle/libtg_oracle_abi.so — the Python host
side (Engine, the drop-in Tetris, TetrisVectorEnv) then drives the CPU restatement through the product's own C ABI."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests import replay  # noqa: E402
from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402

assert N.LIB_PATH.endswith("libtg_oracle_abi.so"), N.LIB_PATH
GOLDEN = os.path.join(ROOT, "tests", "golden")

from tests.test_gpu_parity import EngineBackend, EngineInStepObsBackend, MappedEngineBackend  # noqa: E402

# 1. reference traces through Engine -> C ABI -> restatement (the replays tests/test_gpu_parity.py runs on the GPU)
for name in ("trace_10x20_g1", "clears_10x20", "trace_7x14_q2_g1"):
    replay.replay_trace(EngineInStepObsBackend, replay.load(GOLDEN, name), max_steps=120)
replay.replay_trace(MappedEngineBackend, replay.load(GOLDEN, "mapped_alias_left_10x20"), max_steps=120)
# 2. the hand-made states, uploaded through tg_set_state_dev
fx = replay.load(GOLDEN, "poked_10x20_g0")
be = EngineBackend(8 * int(fx["n_states"]), 10, 20, False, 4)
be.reset(seeds=replay.poked_seeds(fx))


def up(board, rec):
    db, dr, dg = be.e.get_state_device()
    be.e.dev_upload(db.ptr, np.ascontiguousarray(board)); be.e.dev_upload(dr.ptr, np.ascontiguousarray(rec))
    be.e.set_state_device(board=db, rec=dr)


replay.replay_poked(be, fx, upload=up)
# 3. rollouts with trajectories, both rule-sets, against pyoracle's own binding of the same restatement
for rs, obs, pobs, na in ((N.RULESET_NUMPY, N.OBS_BOARD, po.OBS_BOARD, 8), (N.RULESET_FUNCTIONAL, N.OBS_FN, po.OBS_FN, 7)):
    B, K = 64, 24
    e = Engine(B, 10, 20, ruleset=rs, obs_mode=obs, autoreset=N.AUTORESET_SAME_STEP, episode_stats=True)
    o = po.OracleBatch(B, 10, 20, rs)
    seeds = np.arange(3, 3 + B, dtype=np.uint64)
    e.reset(seeds); o.reset(seeds)
    acts = np.random.default_rng(rs).integers(0, na, (K, B)).astype(np.int8)
    tobs, trew, tterm, tlines = e.rollout(acts)
    for k in range(K):
        ro, to, lo = o.step(acts[k].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        assert np.array_equal(trew[k], ro) and np.array_equal(tterm[k], to) and np.array_equal(tlines[k], lo)
        assert np.array_equal(tobs[k], o.observe(pobs))
    assert e.step_kernel_name().startswith("tgo::cpu_oracle")
    e.close()
# 4. the vector adapter's one-call host step (tg_step_fetch_many, TG_BUF_HOST_PACK, episode statistics)
from tetris_gymnasium_amd.vector import TetrisVectorEnv  # noqa: E402

v = TetrisVectorEnv(96, observation="dict", autoreset_mode="same_step", devices=(0, 0))
o = po.OracleBatch(96)
obs, _ = v.reset(seed=9); o.reset(np.arange(9, 9 + 96, dtype=np.uint64))
rng = np.random.default_rng(4)
n_ep = 0
for t in range(150):
    a = rng.integers(0, 8, 96)
    obs, rew, term, trunc, infos = v.step(a)
    ro, to, lo = o.step(a, autoreset=po.AUTORESET_SAME_STEP)
    ref = o.observe(po.OBS_DICT)
    assert all(np.array_equal(obs[k], ref[k]) for k in ref) and np.array_equal(rew, o.last_rewards()) and np.array_equal(term, to.astype(bool))
    assert np.array_equal(infos["lines_cleared"], lo)
    n_ep += int(term.sum())
assert n_ep > 0
v.close()
print("ORACLE_ABI_OK")
