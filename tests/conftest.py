"""pytest configuration: the `gpu` marker and shared helpers' import path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The device-resident tests hand engine buffers to torch tensors inside THIS process, engine first and torch later: that is
# the case the opt-in TG_HIP_RUNTIME=torch exists for (_native._share_torch_hip_runtime; the default is the system runtime
# the engine was built against).  The SONAME check there still applies.
os.environ.setdefault("TG_HIP_RUNTIME", "torch")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
