import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long forms of GPU tests whose cost is host-side LP solves; skipped unless GNNB_RUN_SLOW=1 "
                                       "(GNNB_RUN_SLOW=1 python -m pytest tests -m 'gpu and slow')")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` is what the driver runs, inside a time limit: the `slow` forms stay out of it unless asked for
    if os.environ.get("GNNB_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow form: set GNNB_RUN_SLOW=1 (python -m pytest tests -m 'gpu and slow')")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    # the parity margins the GPU tests reported (tests/margins.py) -> profiles/r06_parity_margins.json
    from tests import margins
    path = margins.dump(ROOT)
    if path:
        print(f"\nparity margins written to {path}")


@pytest.fixture(scope="session")
def shipped_state():
    from tests.common import shipped_state as f
    return f()


@pytest.fixture(scope="session")
def random_state():
    from tests.common import random_state as f
    return f()
