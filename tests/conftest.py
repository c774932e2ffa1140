import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        from pointcloud_stitching_amd import lib
        return lib.load().pcs_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_present():
    return _gpu_present()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Builds oracle/libpcs_oracle.so if missing."""
    from oracle import pcs_oracle
    pcs_oracle.lib()
    return pcs_oracle


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the product has no fallback.
    pass
