import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "retrieval-scaling_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def rsxlib():
    """The product's Python binding with the HIP library built in-tree."""
    import rsx
    if not os.path.exists(os.path.join(PKG, "csrc", "librsx.so")):
        rsx.build()
    rsx.lib()
    return rsx


@pytest.fixture(scope="session")
def gpu(rsxlib):
    """GPU tests fail loudly (never skip, never fall back) when the device or library is missing."""
    n = rsxlib.get_num_gpus()
    assert n > 0, "no HIP device visible: GPU tests must run on an MI355X box"
    return rsxlib
