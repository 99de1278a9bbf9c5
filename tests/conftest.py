import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SCENE_DATA = os.path.join(ROOT, "scene_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


def _gpu_available() -> bool:
    # the kernel driver's device node; avoids initialising any GPU runtime here
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand when a compiler is around."""
    from tests import oracle_binding

    return oracle_binding.load()


@pytest.fixture(scope="session")
def scene_data():
    return SCENE_DATA
