import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oraclelib
    return oraclelib.oracle()


@pytest.fixture(scope="session")
def reference():
    import oraclelib
    ref = oraclelib.reference()
    if ref is None:
        pytest.skip("real reference (/root/reference -> oracle/_ref) not available on this machine")
    return ref


@pytest.fixture(scope="session")
def gpu_ctx():
    """The product path.  Fails (does not skip) when the HIP library or the GPU is missing:
    a -m gpu run must never pass on a fallback."""
    from dj_brdf_amd import djb
    n = djb.device_count()
    assert n > 0, "no HIP device: -m gpu tests need a real MI355X"
    return djb.default_context(0)
