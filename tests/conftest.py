import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real AMD GPU (run with -m gpu on the MI355X box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle.load()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _hip_extension_present(request):
    """-m gpu runs on a box that normally receives the in-tree liblegged_hip.so with the snapshot; if it did not (fresh clone),
    build it once (hipcc is part of the image) rather than failing every test at load time."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import __graft_entry__ as entry
        if not os.path.isfile(entry.HIP_LIB):
            entry.build()
    yield
