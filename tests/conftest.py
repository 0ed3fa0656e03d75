import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """The shared library is a build artefact (git-ignored): compile it when a fresh checkout runs the tests."""
    lib = os.path.join(ROOT, "video2music_amd", "lib", "libamt_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call([os.path.join(ROOT, "video2music_amd", "csrc", "build.sh")])


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load
