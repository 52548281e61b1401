import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress (one unexplained stop of the suite in round 6) ends the run with a
    stack dump after 15 minutes instead of sitting on the box until somebody's outer limit: the slowest test takes ~80 s.  The
    "thread" method: a hang inside a HIP call never returns to the interpreter, a signal handler would not run."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for it in items:
        if it.get_closest_marker("gpu") is not None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(float(os.environ.get("ACX_TEST_TIMEOUT_S", "900")), method="thread"))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def prompts_table():
    import json
    with open(os.path.join(REPO, "anomalyclip_amd", "data", "prompts.json")) as f:
        return json.load(f)
