"""Zero-copy hand-over of engine buffers to torch tensors (DeviceArray: __cuda_array_interface__ / DLPack) — the two cases of
the suite that need torch's HIP runtime and the engine's in ONE process.  NOT collected by the main pytest session (whose
engine runs on the system runtime it was built against: ADVICE r04): tests/test_gpu_torch_interop.py runs this file in
child processes, once per supported load order —
    TG_INTEROP_ORDER=torch_first   torch is imported before the engine (the `"torch" in sys.modules` rule of
                                   _native._share_torch_hip_runtime: nothing to configure)
    TG_INTEROP_ORDER=engine_first  the engine loads first, torch later: needs the opt-in TG_HIP_RUNTIME=torch."""
import os

if os.environ.get("TG_INTEROP_ORDER", "torch_first") == "torch_first":
    import torch  # noqa: F401  (before anything loads libtetris_gym_hip.so)

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402
from tetris_gymnasium_amd.vector import TetrisVectorEnv  # noqa: E402

N.load()  # (engine_first: the library — and with TG_HIP_RUNTIME=torch the bundled runtime — is in before torch is imported below)


class _D2HCounter:
    """Counts every device → host transfer the Python layer can make (tg_fetch, tg_dev_download, tg_get_state)."""

    def __init__(self, monkeypatch):
        self.n = 0
        for name in ("_fetch", "dev_download", "get_state", "fetch_outputs"):
            orig = getattr(Engine, name)

            def wrapped(eng, *a, _orig=orig, **k):
                self.n += 1
                return _orig(eng, *a, **k)

            monkeypatch.setattr(Engine, name, wrapped)


def test_runtime_is_the_one_this_order_promises():
    import torch as _t

    if os.environ.get("TG_INTEROP_ORDER", "torch_first") == "engine_first":
        assert os.environ.get("TG_HIP_RUNTIME") == "torch"
    if _t.cuda.is_available():  # (the bundled runtime was accepted: SONAMEs match on this image)
        assert N.HIP_RUNTIME != "system"


def test_device_array_is_zero_copy_for_torch(monkeypatch):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    B = 4096
    rng = np.random.default_rng(1)
    for obs_mode in (N.OBS_BOARD, N.OBS_DICT, N.OBS_RGB):
        e = Engine(B, 10, 20, obs_mode=obs_mode, autoreset=N.AUTORESET_SAME_STEP, device=0)
        e.reset(np.arange(1, B + 1, dtype=np.uint64))
        for _ in range(25):
            e.step_async(rng.integers(0, 8, B))
        host_obs, host_out = e.observation(), e.fetch_outputs()
        cnt = _D2HCounter(monkeypatch)
        views, out = e.device_array(N.BUF_OBS), e.device_array(N.BUF_STEP_OUT)
        pairs = list(zip(views.values(), host_obs.values())) if isinstance(views, dict) else [(views, host_obs)]
        for v, h in pairs:
            t = torch.as_tensor(v, device="cuda")
            assert t.data_ptr() == v.ptr and tuple(t.shape) == h.shape and cnt.n == 0
            t2 = torch.from_dlpack(v)
            assert t2.data_ptr() == v.ptr and cnt.n == 0
            assert torch.equal(t, t2)
            np.testing.assert_array_equal(t.cpu().numpy(), h)
        for name, h in zip(("reward", "terminated", "lines"), host_out):  # strided views of the 8-byte output records
            t = torch.as_tensor(out[name], device="cuda")
            assert t.data_ptr() == out[name].ptr and cnt.n == 0
            np.testing.assert_array_equal(t.cpu().numpy(), h)
        for name, h in zip(("reward", "terminated", "lines"), host_out):  # ... and their own download path
            np.testing.assert_array_equal(np.asarray(out[name]), h)
        monkeypatch.undo()
        e.close()


def test_vector_env_device_step_and_device_actions():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    B = 3000
    rng = np.random.default_rng(2)
    for kind in ("board", "dict"):
        a = TetrisVectorEnv(B, observation=kind, autoreset_mode="same_step")
        b = TetrisVectorEnv(B, observation=kind, autoreset_mode="same_step")
        a.reset(seed=7); b.reset(seed=7)
        for t in range(40):
            act = rng.integers(0, 8, B)
            obs_h, rew_h, term_h, _, info_h = b.step(act)
            if t % 2:  # actions already on the device (a policy's output): handed over as a device pointer
                ta = torch.as_tensor(act.astype(np.int32), device="cuda")
                torch.cuda.synchronize()
                obs_d, out_d = a.step(ta.data_ptr(), to_host=False)
            else:
                obs_d, out_d = a.step(act, to_host=False)
            r = torch.as_tensor(out_d["reward"], device="cuda").cpu().numpy()
            np.testing.assert_array_equal(r.astype(np.float64), rew_h)
            np.testing.assert_array_equal(torch.as_tensor(out_d["terminated"], device="cuda").cpu().numpy().astype(bool), term_h)
            for k in (obs_h if isinstance(obs_h, dict) else [None]):
                got = torch.as_tensor(obs_d[k] if k else obs_d, device="cuda").cpu().numpy()
                np.testing.assert_array_equal(got, obs_h[k] if k else obs_h)
        a.close(); b.close()


