"""The product's C ABI implemented by the CPU restatement (oracle/tg_oracle_abi.c; SURVEY §8(b): "the identical ABI is implemented by
the CPU restatement"): one binding drives both libraries.  `TG_LIB=oracle/libtg_oracle_abi.so` makes the Python host side — Engine, the
drop-in classes, the vector adapter — run on the oracle, on a box without a GPU: the bisect tool for "is it the kernel or the Python
above the ABI?".  TEST INFRASTRUCTURE: the product never loads it (tests/test_native_abi.py::test_product_never_imports_the_oracle)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libtg_oracle_abi.so")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtg_oracle_abi.so"], stdout=subprocess.DEVNULL)


def test_oracle_abi_exports_every_declared_symbol():
    _build()
    hdr = open(os.path.join(ROOT, "include", "tetris_gym_hip.h")).read()
    declared = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", hdr)) - {"tg_rng_rec"}
    exported = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", LIB]).decode().splitlines() if " T tg_" in l}
    assert declared <= exported, sorted(declared - exported)


def test_python_host_side_runs_on_the_oracle_through_the_c_abi():
    """Reference traces, hand-made states (through tg_set_state_dev), rollouts with trajectories of both rule-sets and the vector
    adapter's one-call step: Engine / TetrisVectorEnv over TG_LIB=libtg_oracle_abi.so, in a child process (the library is chosen at
    import)."""
    _build()
    env = dict(os.environ, TG_LIB=LIB, PYTHONPATH=ROOT, OMP_NUM_THREADS="4")
    env.pop("TG_LIB_IS_CPU_CHECKER", None)
    worker = [sys.executable, os.path.join(ROOT, "tests", "oracle_abi_worker.py")]
    p = subprocess.run(worker, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode != 0 and "not the HIP engine" in p.stderr  # TG_LIB alone is refused: never a silent CPU path
    p = subprocess.run(worker, env=dict(env, TG_LIB_IS_CPU_CHECKER="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ORACLE_ABI_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
