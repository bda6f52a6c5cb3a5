import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the seven golden proofs of the reference's e2e flow as committed fixtures (tests/golden/make_fixtures.py), with the circuit whose
# step programs made them
GOLDEN_PROOFS = [("proof_recursive2_01.json", "recursive2"), ("proof_recursive1_0.json", "recursive1"), ("proof_recursive1_1.json", "recursive1"),
                 ("proof_recursive1_2.json", "recursive1"), ("proof_recursive1_3.json", "recursive1"), ("proof_recursive2_23.json", "recursive2"),
                 ("proof_recursive2_03.json", "recursive2")]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """@pytest.mark.gpu items are skipped (not failed) on a box without a GPU, whether or not -m selected them"""
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no /dev/kfd here)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import gl_oracle
    return gl_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
