"""world_size-2 gloo tests (CPU): the N>1 path of bench.py is correct by construction.

`bench.launch_ranks` — the launcher `python bench.py --gpus N` itself uses — spawns the rank processes (torchrun
environment contract, rendezvous on 127.0.0.1, no HIP call in the parent); the ranks run tests/dist_worker.py, which
drives bench.Dist (barrier, max-over-ranks, per-rank timing gather, contiguous weak-scaling shards with global seeds)
with the CPU oracle standing in for the GPU engine."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def test_bench_launcher_runs_two_ranks(capfd):
    sys.path.insert(0, ROOT)
    import bench

    env_before = dict(os.environ)
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        rc = bench.launch_ranks(2, ["--expect-world", "2"], script=WORKER, timeout_s=240)
    finally:
        os.environ.clear()
        os.environ.update(env_before)
    out = capfd.readouterr().out
    assert rc == 0, out[-2000:]
    assert "DIST_OK" in out


def test_bench_launcher_reports_a_failing_rank():
    sys.path.insert(0, ROOT)
    import bench

    rc = bench.launch_ranks(2, ["--fail-rank", "1"], script=WORKER, timeout_s=240)
    assert rc != 0


def test_bench_cli_is_self_launching_and_checks_world_size():
    """`--gpus N` must agree with a torchrun environment, and the launcher parent must stay free of the HIP library."""
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29534")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("def rank_main")]
    launcher = src[src.index("def launch_ranks"): src.index("class Dist")]
    assert "tetris_gymnasium_amd" not in launcher and "tg_device_count" not in launcher
    assert "import tetris_gymnasium_amd" not in head.split("def run_block")[0]


def test_two_rank_torchrun_job():
    """The driver's own launch line (torch.distributed.run) on the same worker."""
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", WORKER, "--expect-world", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_OK" in p.stdout
