import gzip
import json
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHTS = os.path.join(ROOT, "caro_ai_amd", "data", "weights")  # the reference's shipped checkpoints (package data)
# tests/golden/weights is a link to it; a copy of the tree that lost the link (an archive, a snapshot) gets it back
if not os.path.isdir(os.path.join(GOLDEN, "weights")) and os.path.isdir(WEIGHTS):
    try:
        if os.path.lexists(os.path.join(GOLDEN, "weights")):
            os.remove(os.path.join(GOLDEN, "weights"))
        os.symlink(WEIGHTS, os.path.join(GOLDEN, "weights"))
    except OSError:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden
