"""The driver parses the LAST stdout line of bench.py as one JSON object.  Round 4's line grew to 22 KB and was not parsed
(BENCH_r04.json: parsed null): this runs bench.py's rank path on a stub engine (no GPU, no kernel) with every block of a
default run and checks that the last line is compact, complete and self-consistent, and that the full block set went to
the side file and to prefixed earlier lines."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _StubLib:
    def tg_device_count(self):
        return 1

    def tg_obs_bytes_per_env(self, cfg):
        return 200


class _StubEngine:
    """Engine look-alike: records calls, hands out fake device pointers, reports a plausible stream time."""
    launches = []

    def __init__(self, B, W, H, **kw):
        self.B, self.W, self.H, self.kw = B, W, H, kw
        self.lib, self.cfg = _StubLib(), None
        self._n = 0
        self._ptr = 1 << 20

    def reset(self, seeds):
        assert seeds.shape == (self.B,) and seeds[0] == 1 + self.kw.get("env_offset", 0)

    def dev_alloc(self, n):
        self._ptr += (n + 255) // 256 * 256
        return self._ptr

    def dev_upload(self, d, a):
        assert a.dtype == np.int8 and a.shape[1] == self.B

    def dev_free(self, d):
        pass

    def stream_copy_gbps(self, n, it):
        return 5000.0

    def step_k_device(self, ptr, k):
        self._n += k
        _StubEngine.launches.append(("step_k", k))

    def rollout_device(self, ptr, k, o, r):
        assert o and r and k <= self.kw.get("fuse", 1)
        self._n += k
        _StubEngine.launches.append(("rollout", k))

    def grouped_rollout(self, k, **kw):
        self._n += k

    def sync(self):
        pass

    def timer_start(self):
        self._t0 = self._n

    def timer_stop(self):
        return (self._n - self._t0) * 6.5e-3  # ms: 6.5 us per transition

    def get_state(self):
        z = np.zeros((self.B, 4), np.uint8)
        return z, z, z

    def step_kernel_name(self, fuse=1, sampled=False, traj=False):
        return f"tg::step_kernel_bits<{self.W}, {self.H}, 0, {(1 if fuse > 1 else 0) | (2 if traj else 0) | (8 if sampled else 0)}>"

    def close(self):
        pass


def test_last_line_is_compact_and_complete(monkeypatch, capsys, tmp_path):
    import bench
    import tetris_gymnasium_amd as pkg
    from tetris_gymnasium_amd import _native as N

    monkeypatch.setattr(pkg, "Engine", _StubEngine)
    monkeypatch.setattr(N, "load", lambda: _StubLib())
    side = tmp_path / "blocks.json"
    monkeypatch.setenv("TG_BENCH_BLOCKS_FILE", str(side))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    # the adapter blocks need a GPU: their failure path ({"error": ...}) is part of what the line must survive
    rc = bench.main(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-budget", "0.05"])
    assert rc == 0
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    last = lines[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "hbm_copy_gbps", "shard_checksums", "blocks_file", "blocks"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "u8"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - r["alg_bytes_per_env_step"] * 65536 / (r["kernel_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert "clock" in r and "traffic" in r and "note" not in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert len(d["shard_checksums"]) == 1 and len(d["shard_checksums"][0]) == 16
    # every block of the default run: a triple in the line, the full record in the side file and on its own earlier line
    want = set(bench.block_specs(256, 64)) | {"vector_env", "single_env", "vector_env_shards"}
    assert set(d["blocks"]) == want
    full = json.loads(side.read_text())
    assert set(full["blocks"]) == want and "notes" in full and full["value"] == d["value"]
    tagged = {ln.split(" ", 2)[1] for ln in lines[:-1] if ln.startswith("[bench-block] ")}
    assert tagged == want
    for ln in lines[:-1]:  # nothing but the last line parses as the result object
        assert not ln.lstrip().startswith("{")
    # fused blocks went through tg_rollout, one launch (<= 16 transitions) per call, plus the tg_step_k secondary
    assert ("rollout", 16) in _StubEngine.launches
    assert full["blocks"]["fused_rollout"]["roofline"]["kernel"].endswith(", 3>")
    assert "last_obs_only_us_per_step" in full["blocks"]["fused_rollout"]["roofline"]


def test_global_actions_do_not_depend_on_the_sharding():
    import bench

    whole = bench.global_actions(48, 0, 4096, 8)
    assert whole.dtype == np.int8 and whole.min() == 0 and whole.max() == 7
    for lo, hi in ((0, 2048), (2048, 4096), (1000, 1001)):
        assert np.array_equal(bench.global_actions(48, lo, hi, 8), whole[:, lo:hi])
    counts = np.bincount(whole.ravel(), minlength=8) / whole.size
    assert np.all(np.abs(counts - 0.125) < 0.01)
