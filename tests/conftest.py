import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_available():
    try:
        from recsys2019_deeplearning_evaluation_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must FAIL (not skip) on a GPU box whose native library is broken; without any device they skip."""
    from recsys2019_deeplearning_evaluation_amd import _native
    _native.load()          # raises if libmi355rec.so is missing: no silent fallback
    if not _gpu_available():
        pytest.skip("no HIP device visible")
    return _native.device_name()
