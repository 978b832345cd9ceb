import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """A fresh checkout has no binaries (they are git-ignored): build the HIP library and the
    harness library once per session (a no-op taking ~2 s when they are up to date).  hipcc
    cross-compiles for gfx950 without a GPU."""
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        from cuembed_amd import build
        build.build()
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build(ref=True)
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
