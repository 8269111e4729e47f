"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU build
only — GPU ASan is not available on this pool).  Builds oracle/libtg_oracle_asan.so and replays reference-generated golden
traces, the functional known answers and the grouped fixtures through it in a child process with libasan preloaded;
any sanitizer report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception:
        return None
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_oracle_is_clean_under_asan_and_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("libasan.so not found (gcc -print-file-name=libasan.so)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtg_oracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan, TG_ORACLE_LIB=os.path.join(ROOT, "oracle", "libtg_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               PYTHONDONTWRITEBYTECODE="1")
    sel = ["tests/test_oracle_golden.py::test_oracle_replays_reference_trace[trace_10x20_g1]",
           "tests/test_oracle_golden.py::test_oracle_replays_reference_trace[clears_20x40]",
           "tests/test_oracle_functional.py", "tests/test_oracle_grouped.py", "tests/test_oracle_rng.py", "tests/test_oracle_rgb.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *sel], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=850)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert " passed" in out
