import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("GWEN_HIP_LIB") and os.environ.get("GWEN_ALLOW_VARIANT_TESTS") != "1":
        # an experimental / ablated variant build must never produce a green test run by accident
        # (tools/experiments/k8_variants.sh sets GWEN_ALLOW_VARIANT_TESTS=1 for its A/B parity runs)
        raise pytest.UsageError("GWEN_HIP_LIB is set: the tests run against the product library only (unset it)")


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library; building is cheap and idempotent."""
    from gwen_amd import build as _b
    _b.build()
    from gwen_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def cref():
    from oracle import gcn_ref
    gcn_ref.build()
    return gcn_ref
