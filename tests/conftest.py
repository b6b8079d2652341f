import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# a GPU memory fault reaches the host as a bare SIGABRT: have the library name the last kernel launches when that happens
os.environ.setdefault("RESNET_MI_TRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle
    o = Oracle("f32")
    o.set_threads(min(16, os.cpu_count() or 1))
    return o


@pytest.fixture(scope="session")
def oracle64():
    from oracle.oracle_py import Oracle
    o = Oracle("f64")
    o.set_threads(min(16, os.cpu_count() or 1))
    return o


@pytest.fixture(scope="session")
def ops():
    from resnet_amd.ops import Ops
    o = Ops()
    if o.L.mi_device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box (no CPU fallback exists)")
    return o
