"""pytest configuration: the `gpu` marker and shared helpers' import path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# (No TG_HIP_RUNTIME here: the session runs on the HIP runtime the engine picks by default — the system one it was built
# against.  The two torch zero-copy cases run in child processes: tests/test_gpu_torch_interop.py.)


def pytest_report_header(config):
    mode = os.environ.get("TG_HIP_RUNTIME", "auto")
    return (f"tetris_gymnasium_amd: TG_HIP_RUNTIME={mode} "
            f"({'system libamdhip64 unless torch is already imported' if mode == 'auto' else mode}); "
            f"_native.HIP_RUNTIME is printed per process with TG_VERBOSE=1")


def pytest_terminal_summary(terminalreporter):
    """which HIP runtime the engine of THIS process actually ran on (the log of a GPU run shows what was tested)"""
    try:
        from tetris_gymnasium_amd import _native

        if _native._lib is not None:
            terminalreporter.write_line(f"tetris_gymnasium_amd: engine library {_native.LIB_PATH}, HIP runtime = {_native.HIP_RUNTIME}")
    except Exception:  # pragma: no cover
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
