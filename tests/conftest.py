import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # -m gpu on a GPU-less box: skip instead of failing in zgml_hip_create.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: GPU tests run on the MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def hip_backend():
    from zgml_amd import Backend
    be = Backend(0)
    yield be
    be.close()
