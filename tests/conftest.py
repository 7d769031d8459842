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
