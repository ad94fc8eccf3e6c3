import importlib
import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the frame-loop tests keep 5-6 HIP streams busy: more hardware queues than the default 4 (read when HIP initialises; the package itself only warns)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ct():
    """The product package (its directory name is not a Python identifier)."""
    return importlib.import_module("3deecelltracker_amd")


@pytest.fixture(scope="session")
def golden_dir():
    return REPO / "tests" / "golden"
