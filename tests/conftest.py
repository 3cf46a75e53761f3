import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(HERE, "golden", "golden.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_unpredict():
    return dict(np.load(os.path.join(HERE, "golden", "golden_unpredict.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def rsqrt_table():
    d = np.load(os.path.join(HERE, "golden", "rsqrt_table.npz"), allow_pickle=False)
    return d["tab"].astype(np.float32), int(d["bits"])


@pytest.fixture(scope="session")
def oracle(rsqrt_table):
    """The C restatement, running with the FIXTURE's rsqrtss table (Intel), so results do not
    depend on the CPU the tests run on."""
    from oracle_lib import Oracle
    o = Oracle()
    o.set_rsqrt_table(*rsqrt_table)
    return o
